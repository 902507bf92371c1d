"""No-GPU checks of the boundary and the host logic: the C-ABI library loads and exports every
symbol include/flowdn.h declares (no compute is called), host-only entry points work, volume I/O
round-trips, the CLI parses the reference's options."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "flowdn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fdn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(fdn):
    lib = ctypes.CDLL(fdn._lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"libflowdn.so does not export {name}"
    assert sorted(fdn._lib.EXPORTS) == declared            # the ctypes layer binds exactly the header


from conftest import assert_kernel as _assert_kernel


def test_host_only_entry_points(fdn):
    lib = fdn._lib.load()
    assert b"gfx950" in lib.fdn_version()
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_kernels.npz"))
    for i, s in enumerate(g["sigmas"]):
        _assert_kernel(fdn.get_gaussian_kernel(float(s)), g[f"k{i}"], float(s))
    with pytest.raises(fdn._lib.FlowdnError):
        fdn.get_gaussian_kernel(-1.0)
    # NULL handle is an error with a message, not a crash
    assert lib.fdn_synchronize(None) < 0 and b"NULL" in lib.fdn_last_error()


def test_mean_host_is_numpys_float32_mean(fdn):
    rng = np.random.default_rng(0)
    for n in (1, 5, 8, 100, 128, 129, 1000, 4099, 65537, 1 << 20, (1 << 21) + 77):
        a = (rng.standard_normal(n) * 100 + 1000).astype(np.float32)
        assert fdn._lib.mean_host(a) == a.mean(), n
    v = (rng.standard_normal((13, 57, 91)) * 50 + 300).astype(np.float32)
    assert fdn._lib.mean_host(v) == v.mean()


def test_no_cpu_fallback_when_library_is_missing(tmp_path, monkeypatch):
    import flowdenoising_amd._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "libflowdn.so"))
    with pytest.raises(L.FlowdnError, match="no CPU fallback"):
        L.load()


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under flowdenoising_amd/ (or the CLI script) may
    import, include, link or dlopen it.  (Comments may cite it.)"""
    bad_py = re.compile(r"^\s*(from\s+oracle|import\s+oracle)", re.M)
    bad_c = re.compile(r"#\s*include[^\n]*oracle|libfdn_oracle|fdo_[a-z_]+\s*\(")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "flowdenoising_amd")):
        for fn in files:
            src = open(os.path.join(dirpath, fn), errors="ignore").read() if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")) else ""
            if fn.endswith(".py"):
                assert not bad_py.search(src) and "libfdn_oracle" not in src, fn
            elif src:
                assert not bad_c.search(src), fn
    assert not bad_py.search(open(os.path.join(ROOT, "flowdenoising.py")).read())


def test_mrc_roundtrip_and_modes(tmp_path):
    from flowdenoising_amd import io as fio
    rng = np.random.default_rng(0)
    v = rng.standard_normal((5, 6, 7)).astype(np.float32)
    p = str(tmp_path / "a.mrc")
    fio.write_mrc(p, v)
    assert os.path.getsize(p) == 1024 + v.nbytes
    assert np.array_equal(fio.read_mrc(p), v)
    assert np.array_equal(np.asarray(fio.read_mrc(p, mmap=True)), v)
    head = open(p, "rb").read(1024)
    assert head[208:212] == b"MAP " and np.frombuffer(head[:16], "<i4").tolist() == [7, 6, 5, 2]
    assert abs(np.frombuffer(head[76:88], "<f4")[2] - v.mean()) < 1e-6
    # other modes + extended header + big-endian, written by hand
    for mode, dt in ((0, "i1"), (1, "i2"), (6, "u2")):
        for order in ("<", ">"):
            a = (rng.random((3, 4, 5)) * 100).astype(order + dt)
            h = bytearray(1024)
            h[0:16] = np.array([5, 4, 3, mode], order + "i4").tobytes()
            h[92:96] = np.array([32], order + "i4").tobytes()
            h[208:212] = b"MAP "
            h[212:216] = bytes([0x44, 0x44, 0, 0]) if order == "<" else bytes([0x11, 0x11, 0, 0])
            q = str(tmp_path / f"m{mode}{'le' if order == '<' else 'be'}.mrc")
            open(q, "wb").write(bytes(h) + b"\0" * 32 + a.tobytes())
            got = fio.read_mrc(q)
            assert got.shape == (3, 4, 5) and np.array_equal(got, a)
    open(str(tmp_path / "bad.mrc"), "wb").write(b"\0" * 100)
    with pytest.raises(ValueError):
        fio.read_mrc(str(tmp_path / "bad.mrc"))


def test_tiff_roundtrip(tmp_path):
    from flowdenoising_amd import io as fio
    rng = np.random.default_rng(1)
    for dt in (np.uint8, np.uint16, np.int16, np.float32):
        v = (rng.random((4, 9, 11)) * 200).astype(dt)
        p = str(tmp_path / f"v_{np.dtype(dt).name}.tif")
        fio.write_tiff(p, v)
        got = fio.read_tiff(p)
        assert got.dtype == v.dtype and np.array_equal(got, v)
    with pytest.raises(ValueError):
        open(str(tmp_path / "x.tif"), "wb").write(b"notatiff")
        fio.read_tiff(str(tmp_path / "x.tif"))


def test_output_dtype_rules(tmp_path):
    """seq:558-571: MRC float32; TIFF uint8 if max < 256 else uint16; par:548: float32 TIFF."""
    from flowdenoising_amd import io as fio
    v = np.linspace(0, 200, 2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    fio.write_volume(str(tmp_path / "a.tif"), v)
    assert fio.read_volume(str(tmp_path / "a.tif")).dtype == np.uint8
    fio.write_volume(str(tmp_path / "b.tif"), v * 10)
    assert fio.read_volume(str(tmp_path / "b.tif")).dtype == np.uint16
    fio.write_volume(str(tmp_path / "c.tif"), v, tiff_float32=True)
    assert fio.read_volume(str(tmp_path / "c.tif")).dtype == np.float32
    fio.write_volume(str(tmp_path / "d.mrc"), v)
    assert np.array_equal(fio.read_volume(str(tmp_path / "d.mrc")), v)
    assert fio.is_mrc_input("x.MRCS") and not fio.is_mrc_output("x.mrcs") and fio.is_mrc_output("x.MRC")


def test_tiff_against_tifffile_if_present(tmp_path):
    """Interoperability with tifffile where an interpreter that has it exists (build container)."""
    import subprocess
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import tifffile"], capture_output=True).returncode:
        pytest.skip("no interpreter with tifffile")
    from flowdenoising_amd import io as fio
    v = (np.random.default_rng(2).random((3, 8, 10)) * 4000).astype(np.uint16)
    mine, theirs = str(tmp_path / "mine.tif"), str(tmp_path / "theirs.tif")
    fio.write_tiff(mine, v)
    np.save(str(tmp_path / "v.npy"), v)
    code = (f"import numpy as np, tifffile; v=np.load(r'{tmp_path}/v.npy'); a=tifffile.imread(r'{mine}'); "
            f"assert a.dtype==v.dtype and np.array_equal(a,v); tifffile.imwrite(r'{theirs}', v, byteorder='>')")
    subprocess.run([py, "-c", code], check=True)
    assert np.array_equal(fio.read_tiff(theirs), v)


def test_cli_options_match_the_reference():
    from flowdenoising_amd.cli import build_parser
    p = build_parser()
    a = p.parse_args([])
    assert a.input == "./volume.mrc" and a.output == "./denoised_volume.mrc"         # seq:451-456
    assert tuple(a.sigma) == (2.0, 2.0, 2.0) and a.winsize == 5 and a.verbosity == 0  # seq:460-469
    assert a.levels is None and not a.no_OF and not a.memory_map and not a.recompute_flow
    a = p.parse_args("-i in.tif -o out.MRC -s 1 2 3 -l 2 -w 7 -v 2 -n -m -p 4 --recompute_flow".split())
    assert (a.input, a.output, a.sigma, a.levels, a.winsize, a.verbosity) == ("in.tif", "out.MRC", ["1", "2", "3"], 2, 7, 2)
    assert a.no_OF and a.memory_map and a.recompute_flow and a.number_of_processes == 4
    a = p.parse_args("--use_GPU --use_threads --show_fingerprint".split())             # gpu:597-598 command lines keep parsing
    assert a.use_GPU and a.use_threads and a.show_fingerprint


def test_slab_plan():
    from flowdenoising_amd.distributed import SlabPlan, split
    assert split(10, 3) == [(0, 4), (4, 7), (7, 10)]
    p = SlabPlan((512, 1024, 1024), 8, 3)
    assert (p.z0, p.zlen) == (192, 64)
    assert p.owner(1, 1023) == 7
    with pytest.raises(ValueError):
        SlabPlan((4, 100, 100), 8, 0)


def test_synth_slabs_tile_the_volume():
    from flowdenoising_amd.synth import make_volume
    full = make_volume((12, 20, 24), seed=9, noise=0.0)
    parts = [make_volume((12, 20, 24), seed=9, noise=0.0, z0=z0, zlen=4) for z0 in (0, 4, 8)]
    np.testing.assert_allclose(np.concatenate(parts), full, rtol=1e-6)


def test_sweep_params_layout_matches_the_header(tmp_path):
    """ctypes' SweepParams against include/flowdn.h's struct as gcc lays it out: size and every field offset."""
    import ctypes
    import subprocess
    from flowdenoising_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = [f[0] for f in _lib.SweepParams._fields_]
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "flowdn.h"\nint main(void) { printf("%zu", sizeof(fdn_sweep_params));\n'
           + "".join(f'printf(" %zu", offsetof(fdn_sweep_params, {n}));\n' for n in names) + "return 0; }\n")
    (tmp_path / "l.c").write_text(src)
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), "-o", str(tmp_path / "l"), str(tmp_path / "l.c")])
    got = [int(v) for v in subprocess.check_output([str(tmp_path / "l")]).split()]
    assert got[0] == ctypes.sizeof(_lib.SweepParams)
    assert got[1:] == [getattr(_lib.SweepParams, n).offset for n in names]
    assert (_lib.WARP_F32, _lib.WARP_F64_PADDED, _lib.WARP_ROUND_INT, _lib.WARP_FIXED_U8) == (0, 1, 2, 3)


def test_transport_header_is_c_and_its_structs_match_ctypes(tmp_path):
    """include/flowdn_rccl.h compiles as plain C99 (a C caller includes it) and fdn_msg / fdn_comm have the layout the ctypes
    layer assumes."""
    import ctypes
    import subprocess
    from flowdenoising_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "flowdn_rccl.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %d %d %d", '
           'sizeof(fdn_msg), offsetof(fdn_msg, bytes), offsetof(fdn_msg, peer), offsetof(fdn_msg, is_send), sizeof(fdn_comm), '
           'offsetof(fdn_comm, exchange), offsetof(fdn_comm, allgather_host), FDN_TRANSPORT_RCCL, FDN_TRANSPORT_SHM, FDN_TRANSPORT_NULL); return 0; }\n')
    (tmp_path / "t.c").write_text(src)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-I", os.path.join(root, "include"), "-o", str(tmp_path / "t"), str(tmp_path / "t.c")])
    got = [int(v) for v in subprocess.check_output([str(tmp_path / "t")]).split()]
    assert got[:4] == [ctypes.sizeof(_lib.FdnMsg), _lib.FdnMsg.bytes.offset, _lib.FdnMsg.peer.offset, _lib.FdnMsg.is_send.offset]
    assert got[4:7] == [ctypes.sizeof(_lib.FdnComm), _lib.FdnComm.exchange.offset, _lib.FdnComm.allgather_host.offset]
    assert got[7:] == [_lib.TRANSPORT_RCCL, _lib.TRANSPORT_SHM, _lib.TRANSPORT_NULL]


def test_tiff_deflate_pages_and_predictor(tmp_path):
    """Deflate-compressed TIFF stacks (compression 8, what tifffile / Bio-Formats write with `compress`; the reference reads
    them through skimage.io.imread, seq:517): strips are zlib streams, predictor 2 is a running sum along the row.  Page
    ranges (a multi-GPU rank's slab) and the page directory alone work without decoding the other pages."""
    import struct
    import zlib
    from flowdenoising_amd import io as fio
    rng = np.random.default_rng(1)
    for dt in (np.uint8, np.uint16, np.int16, np.float32):
        v = (rng.random((5, 33, 47)) * 300).astype(dt)
        fio.write_tiff(str(tmp_path / "c.tif"), v, deflate=True)
        assert np.array_equal(fio.read_tiff(str(tmp_path / "c.tif")), v)
        assert np.array_equal(fio.read_tiff(str(tmp_path / "c.tif"), zrange=(1, 4)), v[1:4])
        assert fio.read_tiff(str(tmp_path / "c.tif"), shape_only=True)[0] == v.shape
    v = (rng.random((8, 16)) * 60000).astype(np.uint16)            # one page, two strips, horizontal differencing
    diff = v.copy()
    diff[:, 1:] = v[:, 1:] - v[:, :-1]
    s0, s1 = zlib.compress(diff[:4].tobytes()), zlib.compress(diff[4:].tobytes())
    ent = [(256, 4, 1, 16), (257, 4, 1, 8), (258, 3, 1, 16), (259, 3, 1, 8), (262, 3, 1, 1), (277, 3, 1, 1), (278, 4, 1, 4), (317, 3, 1, 2), (339, 3, 1, 1)]
    n = len(ent) + 2
    ext = 8 + 2 + n * 12 + 4
    d0 = ext + 16
    ent = sorted(ent + [(273, 4, 2, ext), (279, 4, 2, ext + 8)])
    b = b"II" + struct.pack("<HI", 42, 8) + struct.pack("<H", n)
    for tag, typ, cnt, val in ent:
        b += struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<H", val).ljust(4, b"\0") if typ == 3 else struct.pack("<I", val))
    b += struct.pack("<I", 0) + struct.pack("<II", d0, d0 + len(s0)) + struct.pack("<II", len(s0), len(s1)) + s0 + s1
    (tmp_path / "p.tif").write_bytes(b)
    assert np.array_equal(fio.read_tiff(str(tmp_path / "p.tif"))[0], v)
    # LZW stays refused, loudly
    (tmp_path / "l.tif").write_bytes(b.replace(struct.pack("<HHIHH", 259, 3, 1, 8, 0), struct.pack("<HHIHH", 259, 3, 1, 5, 0)))
    with pytest.raises(ValueError, match="compression 5"):
        fio.read_tiff(str(tmp_path / "l.tif"))


def test_volume_writer_slab_by_slab_equals_whole_file_writers(tmp_path):
    """io.VolumeWriter (the CLI files the result slab by slab while later slabs are still coming from the GPU) writes the
    same bytes as write_mrc / write_tiff on the whole array; an MRC header needs the statistics up front."""
    from flowdenoising_amd import io as fio
    rng = np.random.default_rng(0)
    for dt in (np.uint8, np.uint16, np.float32):
        v = (rng.random((7, 9, 11)) * 200).astype(dt)
        fio.write_tiff(str(tmp_path / "a.tif"), v)
        w = fio.VolumeWriter(str(tmp_path / "b.tif"), v.shape, v.dtype)
        for z0 in (0, 3, 5):
            w.write_slab(v[z0:{0: 3, 3: 5, 5: 7}[z0]])
        w.close()
        assert (tmp_path / "a.tif").read_bytes() == (tmp_path / "b.tif").read_bytes()
    v = rng.random((7, 9, 11)).astype(np.float32)
    fio.write_mrc(str(tmp_path / "a.mrc"), v)
    st = fio.volume_stats(v)
    assert st["min"] == v.min() and st["max"] == v.max() and abs(st["mean"] - v.astype(np.float64).mean()) < 1e-15
    w = fio.VolumeWriter(str(tmp_path / "b.mrc"), v.shape, np.float32, st)
    w.write_slab(v[:4])
    w.write_slab(v[4:])
    w.close()
    assert (tmp_path / "a.mrc").read_bytes() == (tmp_path / "b.mrc").read_bytes()
    assert np.array_equal(fio.read_mrc(str(tmp_path / "b.mrc"), mmap=True), v)
    with pytest.raises(ValueError, match="statistics"):
        fio.VolumeWriter(str(tmp_path / "c.mrc"), v.shape, np.float32)


def test_several_writers_fill_one_file_like_a_single_writer(tmp_path):
    """The ranks of `--gpus N` write their own Z-slabs into ONE output file (io.VolumeWriter(z0=..., create=False)): page /
    slice positions follow from the shape alone, so writers that start at different slices, in any order, produce the
    single writer's bytes (cli._run_sharded)."""
    from flowdenoising_amd import io as fio
    rng = np.random.default_rng(1)
    for dt, shape in ((np.uint8, (9, 6, 10)), (np.uint16, (9, 7, 13)), (np.float32, (5, 8, 9))):
        v = (rng.random(shape) * 200).astype(dt)
        fio.write_tiff(str(tmp_path / "a.tif"), v)
        parts = [(0, 2), (2, 6), (6, shape[0])] if shape[0] > 6 else [(0, 1), (1, 4), (4, shape[0])]
        w0 = fio.VolumeWriter(str(tmp_path / "b.tif"), shape, dt, None, z0=0, create=True)       # rank 0 makes the file ...
        for z0, z1 in reversed(parts[1:]):                                                       # ... the others fill theirs in first
            w = fio.VolumeWriter(str(tmp_path / "b.tif"), shape, dt, None, z0=z0, create=False)
            w.write_slab(v[z0:z1])
            w.close()
        w0.write_slab(v[parts[0][0]:parts[0][1]])
        w0.close()
        assert (tmp_path / "a.tif").read_bytes() == (tmp_path / "b.tif").read_bytes()
        assert np.array_equal(fio.read_tiff(str(tmp_path / "b.tif")), v)
    v = rng.random((7, 9, 11)).astype(np.float32)
    st = fio.volume_stats(v)
    fio.write_mrc(str(tmp_path / "a.mrc"), v, stats=st)
    w0 = fio.VolumeWriter(str(tmp_path / "b.mrc"), v.shape, np.float32, st, z0=0, create=True)
    w2 = fio.VolumeWriter(str(tmp_path / "b.mrc"), v.shape, np.float32, st, z0=5, create=False)
    w1 = fio.VolumeWriter(str(tmp_path / "b.mrc"), v.shape, np.float32, st, z0=2, create=False)
    w2.write_slab(v[5:]); w2.close()
    w0.write_slab(v[:2]); w0.close()
    w1.write_slab(v[2:5]); w1.close()
    assert (tmp_path / "a.mrc").read_bytes() == (tmp_path / "b.mrc").read_bytes()


def test_combine_slice_stats_matches_numpy_and_propagates_nan(fdn):
    rng = np.random.default_rng(3)
    v = rng.standard_normal((6, 5, 7))
    rows1 = np.array([[s.min(), s.max(), s.sum(), 0.0] for s in v])
    mean = fdn._lib.combine_slice_stats(rows1, v.size)["mean"]
    rows2 = np.array([[0, 0, 0, ((s - mean) ** 2).sum()] for s in v])
    st = fdn._lib.combine_slice_stats(rows1, v.size, rows2)
    assert st["min"] == v.min() and st["max"] == v.max()
    assert abs(st["mean"] - v.mean()) < 1e-14 and abs(st["std"] - v.std()) < 1e-14
    rows1[3, 0] = rows1[3, 1] = np.nan                 # a slice with a NaN voxel: numpy's min / max propagate it
    st = fdn._lib.combine_slice_stats(rows1, v.size)
    assert np.isnan(st["min"]) and np.isnan(st["max"])


def test_product_does_not_import_torch():
    """north_star: "PyTorch-ROCm is not needed here".  The drop-in (CLI single- and multi-GPU, operators, I/O, the launcher,
    the ctypes layer, streaming) imports no torch at module level, and importing it all leaves torch out of sys.modules;
    torch stays confined to distributed.py's gloo / RCCL rehearsal engine (imported lazily inside its classes) and bench.py."""
    import subprocess
    import sys
    prog = ("import sys; sys.path.insert(0, %r)\n"
            "import flowdenoising_amd, flowdenoising_amd.cli, flowdenoising_amd.launch, flowdenoising_amd.operators, flowdenoising_amd.io, "
            "flowdenoising_amd.streaming, flowdenoising_amd.flower, flowdenoising_amd.distributed, flowdenoising_amd._lib\n"
            "assert 'torch' not in sys.modules, 'torch imported'\nprint('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]
    bad = re.compile(r"^(import torch|from torch)", re.M)
    for fn in ("cli.py", "launch.py", "operators.py", "io.py", "_lib.py", "streaming.py", "flower.py", "synth.py", "__init__.py"):
        assert not bad.search(open(os.path.join(ROOT, "flowdenoising_amd", fn)).read()), fn
    src = open(os.path.join(ROOT, "flowdenoising_amd", "cli.py")).read()
    assert "torch.distributed" not in src.split('"""', 2)[2]        # (the module docstring may say what it no longer does)


def test_mapped_mrc_writer_writes_the_file_write_mrc_writes(tmp_path):
    """io.MappedMrcWriter (the CLI's MRC output: the file mapped, faulted in and -- where allowed -- page-locked while the
    passes run, the result copied into its pages): byte for byte what write_mrc makes of the same array and statistics,
    with and without a handle that can page-lock."""
    from flowdenoising_amd import io as fio

    class NoPin:
        def host_register(self, a):
            return False

        def host_unregister(self, a):
            raise AssertionError("nothing was registered")

    class Pin:
        def __init__(self):
            self.live = 0

        def host_register(self, a):
            assert a.dtype == np.uint8 and a.size == 1024 + 4 * 5 * 33 * 47
            self.live += 1
            return True

        def host_unregister(self, a):
            self.live -= 1

    v = (np.random.default_rng(1).standard_normal((5, 33, 47)) * 10).astype(np.float32)
    st = fio.volume_stats(v)
    fio.write_mrc(str(tmp_path / "ref.mrc"), v, stats=st)
    want = open(tmp_path / "ref.mrc", "rb").read()
    for name, h in (("a.mrc", NoPin()), ("b.mrc", Pin())):
        w = fio.MappedMrcWriter(str(tmp_path / name), v.shape)
        assert w.prepare(h) == isinstance(h, Pin)
        assert w.data.shape == v.shape and w.data.flags["C_CONTIGUOUS"] and w.data.flags["WRITEABLE"]
        w.data[...] = v
        w.finish(st)
        w.close()                                   # a second close is harmless
        assert open(tmp_path / name, "rb").read() == want
        assert getattr(h, "live", 0) == 0
    # a run that fails before finish() leaves an existing output file as it was (the reference writes after its last pass,
    # seq:558-564) and no temporary file behind; until finish() the output path is not touched at all
    before = sorted(os.listdir(tmp_path))
    w = fio.MappedMrcWriter(str(tmp_path / "a.mrc"), v.shape)
    w.prepare(NoPin())
    w.data[...] = 0
    assert open(tmp_path / "a.mrc", "rb").read() == want
    w.close()
    assert open(tmp_path / "a.mrc", "rb").read() == want and sorted(os.listdir(tmp_path)) == before
    w = fio.MappedMrcWriter(str(tmp_path / "new.mrc"), v.shape)
    w.close()
    assert not os.path.exists(tmp_path / "new.mrc") and sorted(os.listdir(tmp_path)) == before
    assert (os.stat(tmp_path / "a.mrc").st_mode & 0o777) == (os.stat(tmp_path / "ref.mrc").st_mode & 0o777)


def test_gpu_tests_that_start_processes_are_marked():
    """conftest orders `gpu_subprocess` tests after the in-process GPU tests (its docstring says why).  The marker is put
    on by hand; this is the lint that a GPU test which starts another GPU process has not been left unmarked."""
    import ast
    import glob
    missing = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "test_*.py"))):
        src = open(path).read()
        lines = src.splitlines()
        module_gpu = re.search(r"^pytestmark = pytest\.mark\.gpu\b", src, re.M) is not None
        for node in ast.parse(src).body:
            if not (isinstance(node, ast.FunctionDef) and node.name.startswith("test_")):
                continue
            first = node.decorator_list[0].lineno - 1 if node.decorator_list else node.lineno - 1
            decorators = "\n".join(lines[first:node.lineno - 1])
            body = "\n".join(lines[node.lineno - 1:node.end_lineno])
            if not (module_gpu or "mark.gpu" in decorators):
                continue
            starts = "subprocess." in body or "launch.spawn" in body or "run_in_fresh_process" in body
            if starts and "mark.gpu_subprocess" not in decorators:
                missing.append(f"{os.path.basename(path)}::{node.name}")
    assert not missing, f"GPU tests that start processes without @pytest.mark.gpu_subprocess: {missing}"


def test_the_command_line_script_keeps_the_parent_of_a_multi_gpu_run_off_the_gpu():
    """flowdenoising.py opens the GPU context in a thread while numpy is imported -- except in the parent of `--gpus N`, which
    must never touch a GPU (its rank processes do), whatever spelling of the option argparse accepts (it takes unambiguous
    prefixes), and never under a launcher's rank variables.  The thread is named, so that the first real handle can wait for it."""
    src = open(os.path.join(ROOT, "flowdenoising.py")).read().split('if __name__ == "__main__":')[0]
    ns = {"__file__": os.path.join(ROOT, "flowdenoising.py"), "__name__": "flowdenoising_under_test"}
    exec(compile(src, "flowdenoising.py", "exec"), ns)
    started = []

    class FakeThread:
        def __init__(self, target=None, args=(), daemon=None, name=None):
            self.args, self.name = args, name

        def start(self):
            started.append((self.name, self.args))

        def join(self, timeout=None):
            pass

    ns["threading"].Thread, real = FakeThread, ns["threading"].Thread
    saved = {k: os.environ.pop(k) for k in ("FDN_RANK", "RANK", "FDN_SYSTEM_ROCM") if k in os.environ}
    try:
        for argv, device in ((["-i", "a.mrc"], 0), (["--device", "3"], 3), (["--dev=2", "-n"], 2),
                             (["--gpus", "2"], None), (["--gpus=4"], None), (["--gpu", "2"], None), (["--gp", "2"], None), (["-h"], None)):
            started.clear()
            ns["_early_start"](argv)
            assert started == ([] if device is None else [(ns["WARM_THREAD_NAME"], (device,))]), (argv, started)
        os.environ["RANK"] = "0"
        started.clear()
        ns["_early_start"](["-i", "a.mrc"])
        assert started == []
    finally:
        ns["threading"].Thread = real
        os.environ.pop("RANK", None)
        os.environ.pop("FDN_SYSTEM_ROCM", None)
        os.environ.update(saved)
    assert ns["WARM_THREAD_NAME"] in open(os.path.join(ROOT, "flowdenoising_amd", "operators.py")).read()
