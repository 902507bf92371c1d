#!/usr/bin/env python3
"""Per-class VALU issue model of a kernel's row-step loops: instruction counts of every pipeline-stage
loop (from hipcc's assembly, as tools/isa_histogram.py finds them) x the cycles one wave-instruction of that opcode holds a
SIMD, MEASURED on this chip by tools/ubench/rates.hip (profiles/r04_valu_rates_256cus.txt: s_memtime cycles at 4 waves per
SIMD -- the residency of the Farneback kernels).  Replaces the flat "x 4 cycles" of rounds 1-3.

usage: valu_model.py <file.s> <mangled-name substring> <rates.txt> [--waves 4] [--json out.json]
                     [--launch-ms T --clock-ghz F --band-steps N]      (measured: for the busy fraction)

Output: per loop (= per stage wave) the modelled SIMD issue cycles per row step, by class; the sum over the four stage waves
a SIMD hosts per round of row steps; and, given the measured launch time, clock and number of band row steps, the fraction of
SIMD issue cycles the VALU stream fills."""
import collections
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_histogram as ih  # noqa: E402


def read_rates(path, waves):
    col = {1: 1, 2: 2, 4: 3}[waves]
    rates = {}
    for ln in open(path):
        if ln.startswith("#") or ln.startswith("class"):
            continue
        m = re.match(r"^(.*?)\s{2,}([\d.]+)\s+([\d.]+)\s+([\d.]+)\s", ln)
        if m:
            rates[m.group(1).strip()] = float(m.group(1 + col))
    return rates


# opcode (suffixes _e32 / _e64 / _dpp / _sdwa stripped) -> row of the rates table
ROW = {
    "v_add_f32": "v_add_f32", "v_sub_f32": "v_sub_f32", "v_subrev_f32": "v_sub_f32", "v_mul_f32": "v_mul_f32",
    "v_fma_f32": "v_fma_f32", "v_fmac_f32": "v_fmac_f32", "v_mac_f32": "v_fmac_f32", "v_mad_f32": "v_fma_f32",
    "v_max_f32": "v_max_f32", "v_min_f32": "v_max_f32", "v_med3_f32": "v_med3_f32",
    "v_pk_add_f32": "v_pk_add_f32", "v_pk_mul_f32": "v_pk_mul_f32", "v_pk_fma_f32": "v_pk_fma_f32",
    "v_add_f64": "v_add_f64", "v_mul_f64": "v_mul_f64", "v_fma_f64": "v_fma_f64", "v_fmac_f64": "v_fmac_f64",
    "v_rcp_f64": "v_rcp_f64", "v_cvt_f64_f32": "v_cvt_f64_f32", "v_cvt_f32_f64": "v_cvt_f32_f64", "v_cvt_f64_u32": "v_cvt_f64_u32",
    "v_cvt_f64_i32": "v_cvt_f64_u32", "v_cvt_i32_f32": "v_cvt_i32_f32", "v_cvt_u32_f32": "v_cvt_i32_f32", "v_cvt_f32_i32": "v_cvt_i32_f32",
    "v_cvt_f32_u32": "v_cvt_i32_f32", "v_cvt_f32_ubyte0": "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1": "v_cvt_f32_ubyte0",
    "v_mov_b32": "v_mov_b32", "v_mov_b64": "v_mov_b64", "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32",
    "v_med3_i32": "v_med3_i32", "v_med3_u32": "v_med3_i32", "v_min_i32": "v_med3_i32", "v_max_i32": "v_med3_i32", "v_min_u32": "v_med3_i32", "v_max_u32": "v_med3_i32",
    "v_add_u32": "v_add_u32", "v_sub_u32": "v_add_u32", "v_subrev_u32": "v_add_u32", "v_add_co_u32": "v_add_u32", "v_addc_co_u32": "v_add_u32",
    "v_and_b32": "v_and_b32", "v_or_b32": "v_and_b32", "v_xor_b32": "v_and_b32", "v_not_b32": "v_and_b32", "v_bfe_u32": "v_mad_u32_u24",
    "v_lshlrev_b32": "v_lshlrev_b32", "v_lshrrev_b32": "v_ashrrev_i32", "v_ashrrev_i32": "v_ashrrev_i32",
    "v_lshlrev_b64": "v_add_f64", "v_mul_u32_u24": "v_mul_u32_u24", "v_mul_i32_i24": "v_mul_u32_u24", "v_mad_u32_u24": "v_mad_u32_u24",
    "v_mad_i32_i24": "v_mad_u32_u24", "v_mul_lo_u32": "v_mad_u32_u24", "v_add_lshl_u32": "v_add_lshl_u32", "v_lshl_add_u32": "v_add_lshl_u32",
    "v_add3_u32": "v_add_lshl_u32", "v_lshl_or_b32": "v_add_lshl_u32", "v_and_or_b32": "v_add_lshl_u32", "v_or3_b32": "v_add_lshl_u32",
    "v_rndne_f32": "v_rndne_f32", "v_floor_f32": "v_floor_f32", "v_fract_f32": "v_floor_f32", "v_trunc_f32": "v_floor_f32",
    "v_readfirstlane_b32": "v_readfirstlane_b32", "v_readlane_b32": "v_readfirstlane_b32", "v_writelane_b32": "v_readfirstlane_b32",
    "v_bfi_b32": "v_add_lshl_u32", "v_perm_b32": "v_add_lshl_u32", "v_alignbit_b32": "v_add_lshl_u32",
}


def cost_of(op, rates, unknown):
    if "_dpp" in op:
        return rates["v_mov_b32_dpp wave_shr:1"], "dpp"
    base = re.sub(r"_(e32|e64|sdwa)$", "", op)
    if base.startswith("v_cndmask_b32"):
        if op.endswith("_e64"):
            return rates["v_cndmask_b32_e64 (sgpr)"], "select (SGPR-pair mask)"
        # a VCC select behind the compare that made its mask: the measured (compare, select) pair minus the compare.  (Sixteen
        # selects in a row on a VCC nobody has written cost 12.7 cycles each in the same table: not a pattern compiled code has.)
        return rates["v_cmp_lt_f32 + v_cndmask (pair)"] - rates["v_cmp_lt_f32 vcc"], "select (VCC mask)"
    if base.startswith("v_cmp"):
        return (rates["v_cmp_lt_f32_e64 sgpr"] if op.endswith("_e64") else rates["v_cmp_lt_f32 vcc"]), "compare"
    row = ROW.get(base)
    if row is None:
        unknown[base] += 1
        return rates["v_add_f64"], "other (priced as the 3.3-cycle class)"
    cls = ("f64" if "f64" in base or base == "v_mov_b64" else "packed f32" if base.startswith("v_pk_") else
           "f32" if "_f32" in base and "cvt" not in base else "convert" if "cvt" in base else "integer / move")
    return rates[row], cls


def main():
    path, name, rates_path = sys.argv[1], sys.argv[2], sys.argv[3]
    opt = dict(zip(sys.argv[4::2], sys.argv[5::2]))
    waves = int(opt.get("--waves", 4))
    rates = read_rates(rates_path, waves)
    full, blocks = ih.parse(path, name)
    succ = ih.cfg(blocks)
    loops = [c for c in ih.sccs(len(blocks), succ) if len(c) > 1 or c[0] in succ[c[0]]]
    loops = [c for c in loops if any(t.startswith("s_barrier") for b in c for t in blocks[b][1])]
    loops.sort(key=lambda c: c[0])
    unknown = collections.Counter()
    out = {"kernel": full, "rates": os.path.basename(rates_path), "waves_per_simd": waves, "loops": []}
    print(f"kernel {full}\nrates: {rates_path} (cycles per wave-instruction per SIMD at {waves} waves per SIMD)\n")
    total_cycles = 0.0
    total_insts = 0
    for c in loops:
        ins = [t.split()[0] for b in c for t in blocks[b][1]]
        nbar = sum(1 for t in ins if t.startswith("s_barrier"))
        valu = [t for t in ins if t.startswith("v_")]
        if len(ins) < 40 or not nbar:
            continue
        by_cls = collections.defaultdict(lambda: [0, 0.0])
        for op in valu:
            cyc, cls = cost_of(op, rates, unknown)
            by_cls[cls][0] += 1
            by_cls[cls][1] += cyc
        n = len(valu) / nbar
        cyc = sum(v[1] for v in by_cls.values()) / nbar
        lds = collections.Counter(t for t in ins if t.startswith("ds_"))
        print(f"loop {blocks[c[0]][0]}: {nbar} row steps per trip; per row step: {n:.0f} VALU instructions = {cyc:.0f} SIMD issue cycles "
              f"({cyc / n:.2f} per instruction)")
        for cls, (k, cy) in sorted(by_cls.items(), key=lambda kv: -kv[1][1]):
            print(f"    {cls:38s} {k / nbar:6.1f} instr  {cy / nbar:7.1f} cycles")
        print(f"    LDS instructions per row step: {sum(lds.values()) / nbar:.1f}  ({', '.join(f'{k} {v / nbar:.1f}' for k, v in lds.most_common(6))})")
        out["loops"].append({"label": blocks[c[0]][0], "row_steps_per_trip": nbar, "valu_per_row_step": n, "issue_cycles_per_row_step": cyc,
                             "by_class": {k: {"instructions": v[0] / nbar, "cycles": v[1] / nbar} for k, v in by_cls.items()}})
        total_cycles += cyc
        total_insts += n
    print(f"\nall stage waves of one band (the mix every SIMD hosts, one wave of each stage on average): {total_insts:.0f} VALU instructions = "
          f"{total_cycles:.0f} SIMD issue cycles per band row step ({total_cycles / max(total_insts, 1):.2f} per instruction; the flat model said 4)")
    out["per_band_row_step"] = {"valu_instructions": total_insts, "issue_cycles": total_cycles}
    if unknown:
        print("opcodes without a measured row (priced at the 3.3-cycle class):", dict(unknown))
    if "--launch-ms" in opt:
        ms, ghz, steps = float(opt["--launch-ms"]), float(opt["--clock-ghz"]), float(opt["--band-steps"])
        avail = ms * 1e-3 * ghz * 1e9 * 1024            # SIMD-cycles of one launch: 256 CUs x 4 SIMDs
        used = total_cycles * steps
        print(f"\nmeasured: {ms} ms per launch at {ghz} GHz = {avail:.3e} SIMD-cycles; {steps:.4g} band row steps x {total_cycles:.0f} = {used:.3e} "
              f"-> VALU issue busy {used / avail:.2f} (upper bound: every loop block counted, the EXEC-masked fallback gathers included)")
        out["measured"] = {"launch_ms": ms, "clock_ghz": ghz, "band_row_steps": steps, "valu_issue_busy": used / avail}
    if "--json" in opt:
        json.dump(out, open(opt["--json"], "w"), indent=1)


if __name__ == "__main__":
    main()
