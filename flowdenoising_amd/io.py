"""Volume I/O for the CLI: MRC2014 and multi-page TIFF, numpy only.

Replaces the reference's calls into `mrcfile` and `skimage.io`/`tifffile`
(src/flowdenoising_sequential.py:508-517 read, 558-571 write; src/flowdenoising.py:466-475,
539-548), which are not installed in the target image.  Volumes are (Z, Y, X) arrays.

MRC: 1024-byte header (nx, ny, nz, mode at words 1-4; nsymbt extended-header bytes at word 24;
'MAP ' + machine stamp at bytes 208-215) followed by the raw array; modes 0/1/2/6/12 =
int8/int16/float32/uint16/float16.  Output is always mode 2 with dmin/dmax/dmean/rms filled in,
like mrcfile.new(...).set_data(float32) (seq:562-564).
TIFF: single-sample pages of uint8/uint16/int16/uint32/float32/float64, uncompressed or deflate-compressed
(compression 8 / 32946, predictor 1 or 2), classic or BigTIFF, either byte order; one IFD per page, or an
ImageJ hyperstack with one IFD and contiguous data.
"""
import os
import struct

import numpy as np

_MRC_MODES = {0: np.int8, 1: np.int16, 2: np.float32, 6: np.uint16, 12: np.float16}


# ---------------------------------------------------------------------------- MRC
def read_mrc(path, mmap=False):
    with open(path, "rb") as f:
        head = f.read(1024)
    if len(head) < 1024:
        raise ValueError(f"{path}: shorter than an MRC header")
    stamp = head[212]
    order = ">" if stamp == 0x11 else "<"
    nx, ny, nz, mode = struct.unpack(order + "4i", head[:16])
    if not (0 < nx < 1 << 20 and 0 < ny < 1 << 20 and 0 < nz < 1 << 20) or mode not in _MRC_MODES:
        # some writers leave the stamp empty: try the other byte order before giving up
        order = "<" if order == ">" else ">"
        nx, ny, nz, mode = struct.unpack(order + "4i", head[:16])
        if not (0 < nx < 1 << 20 and 0 < ny < 1 << 20 and 0 < nz < 1 << 20) or mode not in _MRC_MODES:
            raise ValueError(f"{path}: not an MRC file this reader understands (nx,ny,nz,mode = {nx},{ny},{nz},{mode})")
    nsymbt = struct.unpack(order + "i", head[92:96])[0]
    dt = np.dtype(_MRC_MODES[mode]).newbyteorder(order)
    offset = 1024 + max(nsymbt, 0)
    count = nx * ny * nz
    if os.path.getsize(path) < offset + count * dt.itemsize:
        raise ValueError(f"{path}: truncated ({nz}x{ny}x{nx} mode {mode} needs {offset + count * dt.itemsize} bytes)")
    if mmap:
        return np.memmap(path, dtype=dt, mode="r", offset=offset, shape=(nz, ny, nx))
    with open(path, "rb") as f:
        f.seek(offset)
        data = np.fromfile(f, dtype=dt, count=count)
    return data.reshape(nz, ny, nx)


def _mrc_header(shape, stats):
    nz, ny, nx = shape
    h = bytearray(1024)
    struct.pack_into("<10i", h, 0, nx, ny, nz, 2, 0, 0, 0, nx, ny, nz)
    struct.pack_into("<6f", h, 40, float(nx), float(ny), float(nz), 90.0, 90.0, 90.0)   # cella (1 A voxels), cellb
    struct.pack_into("<3i", h, 64, 1, 2, 3)                                             # mapc, mapr, maps
    struct.pack_into("<3f", h, 76, float(stats["min"]), float(stats["max"]), float(stats["mean"]))
    struct.pack_into("<i", h, 88, 1)            # ispg: a volume
    struct.pack_into("<i", h, 92, 0)            # nsymbt
    struct.pack_into("<i", h, 108, 20140)       # nversion
    h[208:212] = b"MAP "
    h[212:216] = bytes([0x44, 0x44, 0, 0])      # little-endian machine stamp
    struct.pack_into("<f", h, 216, float(stats["std"]))
    label = b"flowdenoising_amd (MI355X)"
    struct.pack_into("<i", h, 220, 1)
    h[224:224 + len(label)] = label
    return h


def volume_stats(vol):
    """min / max / mean / std (the MRC header's dmin, dmax, dmean, rms) of a host array, float64 arithmetic."""
    vol = np.asarray(vol)
    if vol.size < (1 << 27):
        v64 = vol.astype(np.float64, copy=False)
        return {"min": float(vol.min()), "max": float(vol.max()), "mean": float(v64.mean()), "std": float(v64.std())}
    s = sum(float(z.sum(dtype=np.float64)) for z in vol)      # slice-wise to bound memory
    mean = s / vol.size
    var = sum(float(((z.astype(np.float64) - mean) ** 2).sum()) for z in vol) / vol.size
    return {"min": float(vol.min()), "max": float(vol.max()), "mean": mean, "std": float(np.sqrt(var))}


def write_mrc(path, vol, stats=None):
    """stats: {"min", "max", "mean", "std"} of `vol` when the caller has them already (the CLI takes them on the GPU,
    fdn_stats_dev); otherwise they are computed here."""
    vol = np.ascontiguousarray(vol, dtype="<f4")
    with open(path, "wb") as f:
        f.write(_mrc_header(vol.shape, stats if stats is not None else volume_stats(vol)))
        vol.tofile(f)


# ---------------------------------------------------------------------------- TIFF
_TIFF_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 8: "h", 9: "i", 11: "f", 12: "d", 16: "Q", 17: "q", 18: "Q"}


def _read_ifd(f, order, off, big):
    f.seek(off)
    n = struct.unpack(order + ("Q" if big else "H"), f.read(8 if big else 2))[0]
    esz = 20 if big else 12
    raw = f.read(n * esz)
    nxt = struct.unpack(order + ("Q" if big else "I"), f.read(8 if big else 4))[0]
    tags = {}
    for i in range(n):
        e = raw[i * esz:(i + 1) * esz]
        tag, typ = struct.unpack(order + "HH", e[:4])
        cnt = struct.unpack(order + ("Q" if big else "I"), e[4:12] if big else e[4:8])[0]
        val = e[12:20] if big else e[8:12]
        fmt = _TIFF_TYPES.get(typ)
        if fmt is None:
            continue
        size = struct.calcsize("=" + fmt) * cnt
        if size > len(val):
            pos = struct.unpack(order + ("Q" if big else "I"), val)[0]
            here = f.tell()
            f.seek(pos)
            data = f.read(size)
            f.seek(here)
        else:
            data = val[:size]
        if typ == 2:
            tags[tag] = data.rstrip(b"\0").decode("latin-1")
        else:
            vals = struct.unpack(order + fmt * cnt, data)
            tags[tag] = vals if typ != 5 else tuple(vals[i] / max(vals[i + 1], 1) for i in range(0, len(vals), 2))
    return tags, nxt


def _page_dtype(tags, order):
    bps = tags.get(258, (1,))[0]
    fmt = tags.get(339, (1,))[0]
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bps not in (8, 16, 32, 64) or (kind == "f" and bps < 32):
        raise ValueError(f"unsupported TIFF sample format (bits {bps}, format {fmt})")
    return np.dtype(f"{order}{kind}{bps // 8}")


def read_tiff(path, zrange=None, shape_only=False):
    """zrange = (z0, z1): read only those pages (every rank of a multi-GPU run reads its own slab);
    shape_only: return ((Z, Y, X), dtype) from the page directory without touching the pixel data."""
    with open(path, "rb") as f:
        bo = f.read(2)
        order = {b"II": "<", b"MM": ">"}.get(bo)
        if order is None:
            raise ValueError(f"{path}: not a TIFF file")
        magic = struct.unpack(order + "H", f.read(2))[0]
        if magic == 42:
            big = False
            off = struct.unpack(order + "I", f.read(4))[0]
        elif magic == 43:
            big = True
            f.read(4)
            off = struct.unpack(order + "Q", f.read(8))[0]
        else:
            raise ValueError(f"{path}: bad TIFF magic {magic}")
        pages = []
        first = None
        while off:
            tags, off = _read_ifd(f, order, off, big)
            if first is None:
                first = tags
            comp = tags.get(259, (1,))[0]
            if comp not in (1, 8, 32946):
                raise ValueError(f"{path}: TIFF compression {comp} is not supported (uncompressed and deflate are)")
            spp = tags.get(277, (1,))[0]
            if spp != 1 and tags.get(284, (1,))[0] != 2:
                raise ValueError(f"{path}: interleaved multi-sample (RGB) pages are not volumes")
            W, H = tags[256][0], tags[257][0]
            dt = _page_dtype(tags, order)
            offs, cnts = tags[273], tags.get(279)
            if comp != 1:  # deflate (zlib) strips, optionally with the horizontal-differencing predictor: decoded when the page is wanted
                pred = tags.get(317, (1,))[0]
                if spp != 1 or cnts is None or pred not in (1, 2) or (pred == 2 and dt.kind == "f"):
                    raise ValueError(f"{path}: this compressed TIFF layout is not supported (samples {spp}, predictor {pred})")
                pages.append((("deflate", tuple(offs), tuple(cnts), pred), H, W, dt))
                continue
            if spp != 1:   # planar samples (tifffile stores a 3- or 4-slice stack this way): one plane = one slice
                if len(offs) != spp:
                    raise ValueError(f"{path}: planar page with several strips per plane is not supported")
                pages.extend((o, H, W, dt) for o in offs)
            elif len(offs) == 1:
                pages.append((offs[0], H, W, dt))
            else:  # several strips: usable as one block when they are contiguous
                if cnts is None or any(offs[i] + cnts[i] != offs[i + 1] for i in range(len(offs) - 1)):
                    buf = bytearray()
                    for o, c in zip(offs, cnts):
                        f.seek(o)
                        buf += f.read(c)
                    pages.append((np.frombuffer(bytes(buf), dtype=dt).reshape(H, W), H, W, dt))
                else:
                    pages.append((offs[0], H, W, dt))
        if not pages:
            raise ValueError(f"{path}: no image pages")
        _, H, W, dt = pages[0]
        desc = first.get(270, "") if isinstance(first.get(270, ""), str) else ""
        if len(pages) == 1 and desc.startswith("ImageJ=") and "images=" in desc:
            n = int(desc.split("images=")[1].split()[0])   # ImageJ hyperstack: one IFD, contiguous pages
            if shape_only:
                return (n, H, W), dt.newbyteorder("=")
            z0, z1 = (0, n) if zrange is None else (max(0, zrange[0]), min(n, zrange[1]))
            f.seek(pages[0][0] + z0 * H * W * dt.itemsize)
            return np.fromfile(f, dtype=dt, count=(z1 - z0) * H * W).reshape(z1 - z0, H, W)
        for (_, h, w, d) in pages:
            if (h, w, d) != (H, W, dt):
                raise ValueError(f"{path}: pages differ in size or type")
        if shape_only:
            return (len(pages), H, W), dt.newbyteorder("=")
        if zrange is not None:
            pages = pages[max(0, zrange[0]):zrange[1]]
        out = np.empty((len(pages), H, W), dtype=dt.newbyteorder("="))
        for i, (src, h, w, d) in enumerate(pages):
            if isinstance(src, np.ndarray):
                out[i] = src
            elif isinstance(src, tuple):     # deflate strips
                import zlib
                _, offs, cnts, pred = src
                raw = bytearray()
                for o, c in zip(offs, cnts):
                    f.seek(o)
                    raw += zlib.decompress(f.read(c))
                if len(raw) != H * W * dt.itemsize:
                    raise ValueError(f"{path}: page {i} decompresses to {len(raw)} bytes, expected {H * W * dt.itemsize}")
                page = np.frombuffer(bytes(raw), dtype=dt).reshape(H, W)
                if pred == 2:                # horizontal differencing: running sum along the row, modulo the sample width
                    page = np.cumsum(page.astype(dt.newbyteorder("=")), axis=1, dtype=dt.newbyteorder("="))
                out[i] = page
            else:
                f.seek(src)
                if dt.isnative:          # straight into its place (np.fromfile would allocate and copy every page)
                    if f.readinto(memoryview(out[i]).cast("B")) != H * W * dt.itemsize:
                        raise ValueError(f"{path}: page {i} is truncated")
                else:
                    out[i] = np.fromfile(f, dtype=dt, count=H * W).reshape(H, W)
        return out


class _TiffPages:
    """Page after page of an uncompressed multi-page TIFF: one strip and one IFD per page, little-endian; BigTIFF above 4 GiB."""

    def __init__(self, f, shape, dtype, deflate=False, first_page=0):
        """first_page > 0 (uncompressed stacks only): this writer adds pages first_page ... of a file whose header and
        earlier pages somebody else writes -- page positions of an uncompressed stack follow from the shape alone, so the
        ranks of a multi-GPU run write their own Z-slabs into one file side by side."""
        self.deflate = bool(deflate)
        dtype = np.dtype(dtype)
        kind = dtype.kind
        if kind not in "uif" or dtype.itemsize not in (1, 2, 4, 8):
            raise ValueError(f"cannot write dtype {dtype} as TIFF")
        self.f, self.dtype = f, dtype.newbyteorder("<")
        self.Z, self.H, self.W = shape
        self.page_bytes = self.H * self.W * dtype.itemsize
        self.big = self.Z * (self.page_bytes + 512) + 1024 > (1 << 32) - (1 << 25)
        self.fmt_code = {"u": 1, "i": 2, "f": 3}[kind]
        self.desc = ('{"shape": [%d, %d, %d]}' % (self.Z, self.H, self.W)).encode() + b"\0"
        head = b"II" + (struct.pack("<HHHQ", 43, 8, 0, 16) if self.big else struct.pack("<HI", 42, 8))
        self.pos = len(head)
        self.z = 0
        if first_page == 0:
            f.write(head)
        else:
            if self.deflate:
                raise ValueError("pages of a compressed stack cannot be placed ahead of time")
            for _ in range(first_page):          # where page `first_page` starts: the layout arithmetic of write_page
                self.pos = self._layout(self.pos, self.z, self.page_bytes)[2]
                self.z += 1

    def _layout(self, pos, z, page_bytes):
        """(IFD size with its out-of-line description, offset of the pixel data, offset of the next IFD) of page z at `pos`."""
        big = self.big
        n = 10 + (1 if z == 0 else 0)
        ifd_size = (8 + n * 20 + 8) if big else (2 + n * 12 + 4)
        extra = len(self.desc) if z == 0 and len(self.desc) > (8 if big else 4) else 0
        data_off = pos + ifd_size + extra
        data_off += (-data_off) % 16
        next_ifd = data_off + page_bytes if z + 1 < self.Z else 0
        next_ifd += (-next_ifd) % 2
        return ifd_size, data_off, next_ifd

    def write_page(self, page):
        f, big, z, Z, H, W, desc, page_bytes = self.f, self.big, self.z, self.Z, self.H, self.W, self.desc, self.page_bytes
        pos = self.pos
        data = None
        if self.deflate:
            import zlib
            data = zlib.compress(np.ascontiguousarray(page, dtype=self.dtype).tobytes(), 1)
            page_bytes = len(data)
        entries = [(256, 4, W), (257, 4, H), (258, 3, self.dtype.itemsize * 8), (259, 3, 8 if self.deflate else 1), (262, 3, 1)]
        if z == 0:
            entries.append((270, 2, desc))
        entries += [(273, 16 if big else 4, None), (277, 3, 1), (278, 4, H), (279, 16 if big else 4, page_bytes),
                    (339, 3, self.fmt_code)]
        n = len(entries)
        ifd_size, data_off, next_ifd = self._layout(pos, z, page_bytes)
        extra = len(desc) if z == 0 and len(desc) > (8 if big else 4) else 0
        assert n == 10 + (1 if z == 0 else 0)            # _layout counts the same entries
        buf = bytearray(struct.pack("<Q" if big else "<H", n))
        for tag, typ, val in entries:
            if tag == 273:
                val = data_off
            if typ == 2:
                cnt = len(val)
                if cnt > (8 if big else 4):
                    field = struct.pack("<Q" if big else "<I", pos + ifd_size)
                else:
                    field = val.ljust(8 if big else 4, b"\0")
            else:
                cnt = 1
                field = struct.pack("<" + {3: "H", 4: "I", 16: "Q"}[typ], val).ljust(8 if big else 4, b"\0")
            buf += struct.pack("<HH", tag, typ) + struct.pack("<Q" if big else "<I", cnt) + field
        buf += struct.pack("<Q" if big else "<I", next_ifd)
        if extra:
            buf += desc
        f.seek(pos)
        f.write(buf)
        f.seek(data_off)
        if data is not None:
            f.write(data)
        else:
            np.ascontiguousarray(page, dtype=self.dtype).tofile(f)
        self.pos = next_ifd
        self.z += 1


def write_tiff(path, vol, deflate=False):
    """One strip and one IFD per page, little-endian; BigTIFF above 4 GiB.  deflate: zlib-compressed strips (compression 8;
    the CLI writes uncompressed pages like the reference's imsave)."""
    vol = np.asarray(vol)
    if vol.ndim == 2:
        vol = vol[None]
    with open(path, "wb") as f:
        pages = _TiffPages(f, vol.shape, vol.dtype, deflate=deflate)
        for z in range(vol.shape[0]):
            pages.write_page(vol[z])
    return path


class VolumeWriter:
    """The output file of the CLI written slab after slab, in Z order, while later slabs are still on their way from the
    GPU (seq:558-571's rules: MRC float32 with header statistics, else a TIFF stack of the array's dtype).
    `stats` (min / max / mean / std of the whole volume) is needed up front for an MRC header."""

    def __init__(self, path, shape, dtype, stats=None, z0=0, create=True):
        """z0 / create: a multi-GPU run writes ONE file from all its ranks -- rank 0 creates it (create=True, z0 = 0: the
        header), the others open the existing file (create=False) and write the slices from their z0 on at the byte
        offsets those slices have in the single-writer file (MRC: 1024 + z0 * Y * X * 4; TIFF: _TiffPages(first_page))."""
        self.mrc = is_mrc_output(path)
        self.f = open(path, "wb" if create else "r+b")
        self.shape = tuple(shape)
        if self.mrc:
            if z0 == 0:
                if stats is None:
                    raise ValueError("an MRC header needs the volume's statistics")
                self.f.write(_mrc_header(self.shape, stats))
            else:
                self.f.seek(1024 + z0 * self.shape[1] * self.shape[2] * 4)
        else:
            self.pages = _TiffPages(self.f, self.shape, dtype, first_page=z0)

    def write_slab(self, slab):
        if self.mrc:
            np.ascontiguousarray(slab, dtype="<f4").tofile(self.f)
        else:
            for page in slab:
                self.pages.write_page(page)

    def close(self):
        self.f.close()


class MappedMrcWriter:
    """The output MRC as a shared mapping of the file itself, made ready WHILE the passes run: the file is created at its
    final size, mapped, and its pages are faulted in and -- where the kernel allows it (tmpfs, hugetlbfs; not the
    dirty-tracked page cache of ext4 / xfs since Linux 6.5) -- page-locked for DMA, so that the result goes from the GPU
    straight into the page cache in one copy at PCIe speed.  What the slab writer (VolumeWriter) pays after the last pass --
    2 GiB of first-touch page faults plus a host copy, 0.4-0.6 s at configs[2] -- then hides behind the kernels.
    seq:558-564 is what this replaces (mrcfile.new + set_data: header statistics, float32 data)."""

    def __init__(self, path, shape):
        """The mapping is of a TEMPORARY file next to `path` (same directory: same file system, so the final rename is atomic
        and the page cache pages stay the ones the GPU wrote): the reference writes its output only after the last pass
        (seq:558-564), so a run that fails on the way must leave an existing output file as it was -- finish() renames
        the temporary file over `path`, close() without finish() removes it."""
        import mmap
        import tempfile
        self.shape = tuple(int(v) for v in shape)
        self.nbytes = 1024 + 4 * int(np.prod(self.shape))
        self.path = str(path)
        d, base = os.path.split(os.path.abspath(self.path))
        fd, self.tmp_path = tempfile.mkstemp(prefix="." + base + ".", suffix=".part", dir=d)
        self.f = os.fdopen(fd, "w+b")
        try:
            # (mkstemp creates mode 0600; an output file is an ordinary one: what open(path, "wb") would have given it)
            umask = os.umask(0)
            os.umask(umask)
            os.chmod(self.tmp_path, 0o666 & ~umask)
            self.f.truncate(self.nbytes)
            self.mm = mmap.mmap(self.f.fileno(), self.nbytes)
        except BaseException:
            self.f.close()
            os.unlink(self.tmp_path)
            raise
        self.finished = False
        self.data = np.frombuffer(self.mm, dtype="<f4", count=int(np.prod(self.shape)), offset=1024).reshape(self.shape)
        self.whole = np.frombuffer(self.mm, dtype=np.uint8)
        self.pinned_by = None

    def prepare(self, handle):
        """Fault the pages in; page-lock them if that is allowed here (returns True then).  Runs in a thread of its own."""
        if handle.host_register(self.whole):
            self.pinned_by = handle
            return True
        try:
            self.mm.madvise(23)                      # MADV_POPULATE_WRITE (Linux 5.14): fault every page in, writable
        except (OSError, ValueError, AttributeError):
            step = 64 << 20
            for off in range(0, self.nbytes, step):  # touch every page
                self.whole[off:off + step:4096] = 0
        return False

    def finish(self, stats):
        self.mm[0:1024] = bytes(_mrc_header(self.shape, stats))
        self.finished = True
        self.close()

    def close(self):
        if self.pinned_by is not None:
            self.pinned_by.host_unregister(self.whole)
            self.pinned_by = None
        if self.mm is not None:
            self.data = self.whole = None
            try:
                self.mm.close()
            except BufferError:                      # a view is still alive somewhere: the mapping goes with it
                pass
            self.mm = None
            self.f.close()
            if self.finished:
                os.replace(self.tmp_path, self.path)     # the output file appears complete, or not at all
            else:
                try:
                    os.unlink(self.tmp_path)
                except OSError:
                    pass


# ---------------------------------------------------------------------------- dispatch (CLI rules)
def is_mrc_input(path):
    """par:466: 'mrc' in the last suffix, case-insensitive (seq:508 accepts only mrc/MRC)."""
    return "mrc" in str(path).split(".")[-1].lower()


def is_mrc_output(path):
    """seq:558 / par:539: suffix exactly 'mrc' or 'MRC'."""
    return str(path).split(".")[-1] in ("MRC", "mrc")


def read_volume(path, mmap=False):
    if is_mrc_input(path):
        return read_mrc(path, mmap=mmap)
    return read_tiff(path)


def volume_info(path):
    """((Z, Y, X), dtype) of a volume file from its header / page directory only."""
    if is_mrc_input(path):
        v = read_mrc(path, mmap=True)
        return tuple(v.shape), v.dtype.newbyteorder("=")
    return read_tiff(path, shape_only=True)


def read_slab(path, z0, z1):
    """Slices [z0, z1) of a volume file, reading only their bytes (one rank's share of a multi-GPU run)."""
    if is_mrc_input(path):
        return np.array(read_mrc(path, mmap=True)[z0:z1])
    return read_tiff(path, zrange=(z0, z1))


def write_volume(path, vol, tiff_float32=False, stats=None):
    """seq:558-571: MRC float32, else TIFF uint8 if max < 256 else uint16 (values truncated by astype);
    tiff_float32=True gives par:548's float32 TIFF.  stats: min / max / mean / std of `vol` if already known."""
    if is_mrc_output(path):
        write_mrc(path, vol.astype(np.float32, copy=False), stats=stats)
    elif tiff_float32:
        write_tiff(path, vol.astype(np.float32, copy=False))
    elif vol.dtype in (np.uint8, np.uint16):            # already downcast (on the GPU, operators.filter_3d_own_mean)
        write_tiff(path, vol)
    elif (stats["max"] if stats is not None else np.max(vol)) < 256:
        write_tiff(path, vol.astype(np.uint8))
    else:
        write_tiff(path, vol.astype(np.uint16))
