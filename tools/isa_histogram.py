#!/usr/bin/env python3
"""Per-loop instruction histogram of one kernel from hipcc's --save-temps assembly (give the
fused kernel's VALU lane-instructions per pixel an address).

usage: isa_histogram.py <file.s> <mangled-name substring> [--blocks]

The kernel's basic blocks are read from the .s file, the control-flow graph is rebuilt from the branch
instructions, and every strongly connected component that contains an s_barrier (= a row-step loop of one pipeline
stage) is reported: instructions by class, summed over the blocks of the loop.  Blocks that hold global loads of the
R1 fallback gather, the y == 0 initialisation or the below-the-image case are inside the same loops; --blocks lists
every block so that the hot path can be told from them."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if "_dpp" in op or op.startswith("v_mov_b32_dpp"):
            return "valu_dpp"
        if re.search(r"_f64|_b64|f64_", op):
            return "valu_f64"
        if op.startswith("v_pk_"):
            return "valu_pk_f32"
        if op.startswith("v_cndmask") or op.startswith("v_cmp"):
            return "valu_cmp_sel"
        if re.search(r"_f32", op):
            return "valu_f32"
        return "valu_int_mov"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "vmem_rd"
    if op.startswith("global_store") or op.startswith("buffer_store"):
        return "vmem_wr"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse(path, name):
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if ln.startswith("_Z") and name in ln.split(":")[0] and ln.rstrip().split(";")[0].strip().endswith(":"):
            start = i
            break
    if start is None:
        sys.exit(f"kernel {name} not found")
    blocks = [["entry", []]]
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks.append([m.group(1), []])
            continue
        t = ln.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        blocks[-1][1].append(t)
    return lines[start].split(":")[0], blocks


def cfg(blocks):
    idx = {b[0]: i for i, b in enumerate(blocks)}
    succ = collections.defaultdict(set)
    for i, (name, ins) in enumerate(blocks):
        fall = True
        for t in ins:
            op = t.split()[0]
            if op == "s_branch":
                succ[i].add(idx[t.split()[1]])
                fall = False
            elif op.startswith("s_cbranch"):
                succ[i].add(idx[t.split()[1]])
            elif op == "s_endpgm":
                fall = False
        if fall and i + 1 < len(blocks):
            succ[i].add(i + 1)
    return succ


def sccs(n, succ):
    index, low, on, st, out, cnt = {}, {}, set(), [], [], [0]
    sys.setrecursionlimit(100000)

    def go(v):
        index[v] = low[v] = cnt[0]
        cnt[0] += 1
        st.append(v)
        on.add(v)
        for w in succ[v]:
            if w not in index:
                go(w)
                low[v] = min(low[v], low[w])
            elif w in on:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            comp = []
            while True:
                w = st.pop()
                on.discard(w)
                comp.append(w)
                if w == v:
                    break
            out.append(sorted(comp))
    for v in range(n):
        if v not in index:
            go(v)
    return out


def hist(ins):
    h = collections.Counter()
    for t in ins:
        h[classify(t.split()[0])] += 1
    return h


def main():
    path, name = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    full, blocks = parse(path, name)
    succ = cfg(blocks)
    print(f"kernel {full}\n{len(blocks)} basic blocks, {sum(len(b[1]) for b in blocks)} instructions")
    loops = [c for c in sccs(len(blocks), succ) if len(c) > 1 or c[0] in succ[c[0]]]
    loops = [c for c in loops if any(t.startswith("s_barrier") for b in c for t in blocks[b][1])]
    loops.sort(key=lambda c: c[0])
    classes = ["valu_f32", "valu_pk_f32", "valu_f64", "valu_dpp", "valu_cmp_sel", "valu_int_mov", "lds", "vmem_rd", "vmem_wr", "salu",
               "waitcnt", "barrier", "branch", "other"]
    for c in loops:
        ins = [t for b in c for t in blocks[b][1]]
        h = hist(ins)
        nb = h["barrier"]
        if len(ins) < 40:
            continue
        valu = sum(v for k, v in h.items() if k.startswith("valu"))
        print(f"\nloop {blocks[c[0]][0]} .. {blocks[c[-1]][0]}: {len(c)} blocks, {len(ins)} instructions, {nb} barriers (row steps per trip), "
              f"VALU {valu} = {valu / max(nb, 1):.0f} per row step")
        print("  " + "  ".join(f"{k} {h[k]}" for k in classes if h[k]))
        ops = collections.Counter(t.split()[0] for t in ins if t.startswith("v_") or t.startswith("ds_") or t.startswith("global_"))
        print("  top opcodes: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(28)))
        if show_blocks:
            for b in c:
                hb = hist(blocks[b][1])
                print(f"    {blocks[b][0]:12s} {len(blocks[b][1]):4d}  " + " ".join(f"{k}:{hb[k]}" for k in classes if hb[k]) + f"  -> {[blocks[s][0] for s in sorted(succ[b])]}")


if __name__ == "__main__":
    main()
