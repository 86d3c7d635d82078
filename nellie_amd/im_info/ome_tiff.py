"""
Minimal OME-BigTIFF writer + memory-mapped reader (no tifffile / ome_types dependency).

Why it exists: Nellie's intermediates are uncompressed, contiguous OME-BigTIFF files that every stage opens
with `tifffile.memmap` (nellie/im_info/verifier.py:967-1070): `im_preprocessed` (float32) and
`im_instance_label` (int32) must appear in that form under `nellie_output/nellie_necessities/`.  Inside a Nellie
installation the drop-in stages use Nellie's own `ImInfo` (tifffile); this module lets them run where tifffile
is absent (the build / GPU image) and produces files laid out the way `tifffile.imwrite(path, shape=..., dtype=...,
bigtiff=True, metadata={"axes": ...}, photometric="minisblack")` lays them out:

    16-byte BigTIFF header | all planes back to back (C order, little endian) | one IFD per (Y, X) plane

with the OME-XML (axes, sizes, PhysicalSizeX/Y/Z, TimeIncrement, pixel type, description) in the first page's
ImageDescription and a <TiffData IFD="0" PlaneCount="N"/> element, so the planes are contiguous and the file is
memory-mappable at one offset.  NOT verified against tifffile in this image (it is not installed): the reader
below and the TIFF 6.0 / BigTIFF / OME-TIFF specifications are the references; see DESIGN.md.
"""
from __future__ import annotations

import os
import re
import struct
import uuid
from xml.sax.saxutils import escape

import numpy as np

_OME_TYPES = {"uint8": "uint8", "int8": "int8", "uint16": "uint16", "int16": "int16", "uint32": "uint32",
              "int32": "int32", "float32": "float", "float64": "double"}
_SAMPLE_FORMAT = {"u": 1, "i": 2, "f": 3}
_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 16: 8, 17: 8, 18: 8, 13: 4}


def ome_xml(shape_tzyx, dtype, dim_res, description="", name="image"):
    """OME-XML for a (T, Z, Y, X) stack; DimensionOrder XYZTC = planes ordered Z fastest, then T."""
    t, z, y, x = (int(s) for s in shape_tzyx)
    dt = np.dtype(dtype)
    attrs = [f'DimensionOrder="XYZTC"', f'Type="{_OME_TYPES[dt.name]}"', f'SizeX="{x}"', f'SizeY="{y}"',
             f'SizeZ="{z}"', 'SizeC="1"', f'SizeT="{t}"', 'BigEndian="false"']
    for ax in ("X", "Y", "Z"):
        v = (dim_res or {}).get(ax)
        if v is not None:
            attrs.append(f'PhysicalSize{ax}="{float(v)!r}"')
            attrs.append(f'PhysicalSize{ax}Unit="µm"')
    if (dim_res or {}).get("T") is not None:
        attrs.append(f'TimeIncrement="{float(dim_res["T"])!r}"')
        attrs.append('TimeIncrementUnit="s"')
    return ('<?xml version="1.0" encoding="UTF-8"?>'
            '<OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06" '
            'xmlns:xsi="http://www.w3.org/2001/XMLSchema-instance" '
            'xsi:schemaLocation="http://www.openmicroscopy.org/Schemas/OME/2016-06 '
            'http://www.openmicroscopy.org/Schemas/OME/2016-06/ome.xsd" Creator="nellie_amd">'
            f'<Image ID="Image:0" Name="{escape(name)}"><Description>{escape(description)}</Description>'
            f'<Pixels ID="Pixels:0" {" ".join(attrs)}>'
            '<Channel ID="Channel:0:0" SamplesPerPixel="1"><LightPath/></Channel>'
            f'<TiffData IFD="0" PlaneCount="{t * z}"/></Pixels></Image></OME>')


def _ifd(entries, next_offset, base):
    """Serialise one BigTIFF IFD placed at file offset `base`; out-of-line values follow the entry table."""
    entries = sorted(entries, key=lambda e: e[0])
    n = len(entries)
    table_size = 8 + 20 * n + 8
    extra = b""
    body = struct.pack("<Q", n)
    for tag, typ, count, value in entries:
        if typ == 2:                                   # ASCII, NUL terminated
            raw = value
        elif typ == 3:
            raw = struct.pack(f"<{count}H", *value)
        elif typ == 4:
            raw = struct.pack(f"<{count}I", *value)
        elif typ == 5:
            raw = struct.pack(f"<{2 * count}I", *value)
        elif typ == 16:
            raw = struct.pack(f"<{count}Q", *value)
        else:
            raise ValueError(typ)
        if len(raw) <= 8:
            field = raw.ljust(8, b"\0")
        else:
            field = struct.pack("<Q", base + table_size + len(extra))
            extra += raw + (b"\0" if len(raw) % 2 else b"")
        body += struct.pack("<HHQ", tag, typ, count) + field
    body += struct.pack("<Q", next_offset)
    return body + extra


def create(path, shape_tzyx, dtype, dim_res=None, description="", data=None):
    """
    Create (or overwrite) an OME-BigTIFF holding a (T, Z, Y, X) stack of `dtype`.  Zero-filled unless `data`
    (same shape) is given.  Returns the byte offset of the first sample.
    """
    t, z, y, x = (int(s) for s in shape_tzyx)
    dt = np.dtype(dtype).newbyteorder("<")
    if dt.name not in _OME_TYPES:
        raise ValueError(f"unsupported dtype {dt}")
    nplanes = t * z
    plane_bytes = y * x * dt.itemsize
    data_offset = 16
    data_bytes = nplanes * plane_bytes
    xml = ome_xml(shape_tzyx, dt, dim_res, description, os.path.basename(path)).encode("utf-8") + b"\0"
    software = b"nellie_amd\0"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    # written under a temporary name and moved into place: a reader (another rank of a multi-process run, a viewer) sees the old
    # complete file or the new complete file, never a truncated one, and a map of the old file keeps its pages
    final_path, path = path, f"{path}.tmp{os.getpid()}_{_host_tag()}_{uuid.uuid4().hex[:8]}"
    _sweep_stale_tmp(final_path)
    try:
        _write_file(path, t, z, y, x, dt, nplanes, plane_bytes, data_offset, data_bytes, xml, software, data)
        os.replace(path, final_path)
    except BaseException:
        try:
            os.remove(path)
        except OSError:
            pass
        raise
    return data_offset


def _host_tag() -> str:
    """This host (and PID namespace, as far as a name can tell) in a form that fits a file name: a process id in a temporary's name
    only means something to the host that wrote it."""
    import hashlib
    import socket
    return hashlib.sha1(socket.gethostname().encode("utf-8", "replace")).hexdigest()[:8]


def _sweep_stale_tmp(final_path, max_age_s=3600.0):
    """A process killed inside create() leaves `<path>.tmp<pid>_<host>_<hex>` behind (possibly a full-size sparse file next to the
    outputs): the next create() of the same path removes such siblings once their writer is gone -- pid not alive, asked only when
    the temporary was written on THIS host (on a shared file system another node's live writer has a pid that means
    nothing here) -- or once they are an hour old, whoever wrote them (also the round-5 names without a host tag)."""
    import glob
    import re
    import time
    here = _host_tag()
    for p in glob.glob(glob.escape(final_path) + ".tmp*"):
        m = re.fullmatch(r"\.tmp(\d+)_(?:([0-9a-f]{8})_)?[0-9a-f]{8}", p[len(final_path):])
        if not m:
            continue
        try:
            alive = True
            if m.group(2) == here:
                try:
                    os.kill(int(m.group(1)), 0)
                except ProcessLookupError:
                    alive = False
                except PermissionError:
                    pass
            if not alive or time.time() - os.path.getmtime(p) > max_age_s:
                os.remove(p)
        except OSError:
            pass


def _write_file(path, t, z, y, x, dt, nplanes, plane_bytes, data_offset, data_bytes, xml, software, data):
    with open(path, "wb") as f:
        first_ifd = data_offset + data_bytes
        first_ifd += first_ifd % 2
        f.write(b"II" + struct.pack("<HHHQ", 43, 8, 0, first_ifd))
        if data is None:
            f.truncate(first_ifd)                              # sparse zeros
        else:
            a = np.ascontiguousarray(data, dtype=dt)
            if a.shape != (t, z, y, x):
                raise ValueError(f"data shape {a.shape} != {(t, z, y, x)}")
            f.write(a.tobytes())
        f.seek(first_ifd)
        pos = first_ifd
        for p in range(nplanes):
            entries = [
                (256, 4, 1, (x,)), (257, 4, 1, (y,)), (258, 3, 1, (dt.itemsize * 8,)), (259, 3, 1, (1,)),
                (262, 3, 1, (1,)), (273, 16, 1, (data_offset + p * plane_bytes,)), (277, 3, 1, (1,)),
                (278, 4, 1, (y,)), (279, 16, 1, (plane_bytes,)), (282, 5, 1, (1, 1)), (283, 5, 1, (1, 1)),
                (296, 3, 1, (1,)), (339, 3, 1, (_SAMPLE_FORMAT[dt.kind],)),
            ]
            if p == 0:
                entries += [(270, 2, len(xml), xml), (305, 2, len(software), software)]
            blob = _ifd(entries, 0, pos)
            nxt = pos + len(blob)
            nxt += nxt % 2
            if p + 1 < nplanes:                                 # patch the next-IFD pointer
                n = len(entries)
                blob = blob[:8 + 20 * n] + struct.pack("<Q", nxt) + blob[8 + 20 * n + 8:]
            f.write(blob)
            if nxt > pos + len(blob):
                f.write(b"\0")
            pos = nxt


class TiffLayout:
    """What the reader found: contiguous sample block + OME metadata (when present)."""

    def __init__(self, offset, dtype, shape, axes, dim_res, description):
        self.offset, self.dtype, self.shape, self.axes = offset, dtype, shape, axes
        self.dim_res, self.description = dim_res, description


def _read_ifds(f, bigtiff, endian):
    if bigtiff:
        f.seek(8)
        off, = struct.unpack(endian + "Q", f.read(8))
    else:
        f.seek(4)
        off, = struct.unpack(endian + "I", f.read(4))
    pages = []
    while off:
        f.seek(off)
        if bigtiff:
            n, = struct.unpack(endian + "Q", f.read(8))
            raw = f.read(20 * n)
            nxt, = struct.unpack(endian + "Q", f.read(8))
            esz, fmt, inl = 20, endian + "HHQ", 8
        else:
            n, = struct.unpack(endian + "H", f.read(2))
            raw = f.read(12 * n)
            nxt, = struct.unpack(endian + "I", f.read(4))
            esz, fmt, inl = 12, endian + "HHI", 4
        tags = {}
        for k in range(n):
            e = raw[k * esz:(k + 1) * esz]
            tag, typ, count = struct.unpack(fmt, e[:esz - inl])
            size = _TYPE_SIZE.get(typ, 1) * count
            if size <= inl:
                val = e[esz - inl:esz - inl + size]
            else:
                ptr, = struct.unpack(endian + ("Q" if bigtiff else "I"), e[esz - inl:])
                here = f.tell()
                f.seek(ptr)
                val = f.read(size)
                f.seek(here)
            tags[tag] = (typ, count, val)
        pages.append(tags)
        off = nxt
    return pages


def _ints(entry, endian):
    typ, count, val = entry
    code = {1: "B", 3: "H", 4: "I", 16: "Q", 13: "I", 17: "q", 18: "Q"}[typ]
    return struct.unpack(endian + f"{count}{code}", val)


def read_layout(path) -> TiffLayout:
    """Parse an uncompressed TIFF / BigTIFF whose planes are stored back to back."""
    with open(path, "rb") as f:
        head = f.read(4)
        endian = {b"II": "<", b"MM": ">"}.get(head[:2])
        if endian is None:
            raise ValueError(f"{path}: not a TIFF file")
        magic, = struct.unpack(endian + "H", head[2:4])
        if magic not in (42, 43):
            raise ValueError(f"{path}: bad TIFF magic {magic}")
        pages = _read_ifds(f, magic == 43, endian)
    if not pages:
        raise ValueError(f"{path}: no image pages")
    p0 = pages[0]
    if not all(t in p0 for t in (256, 257, 258, 273, 279)):
        raise ValueError(f"{path}: first page lacks the baseline TIFF tags")
    if _ints(p0.get(259, (3, 1, struct.pack(endian + "H", 1))), endian)[0] != 1:
        raise ValueError(f"{path}: compressed TIFFs cannot be memory-mapped")
    x = _ints(p0[256], endian)[0]
    y = _ints(p0[257], endian)[0]
    bits = _ints(p0[258], endian)[0]
    fmt = _ints(p0.get(339, (3, 1, struct.pack(endian + "H", 1))), endian)[0]
    kind = {1: "u", 2: "i", 3: "f"}[fmt]
    dtype = np.dtype(f"{endian}{kind}{bits // 8}")
    plane_bytes = x * y * dtype.itemsize
    offsets = []
    for pg in pages:
        so = _ints(pg[273], endian)
        sc = _ints(pg[279], endian)
        if any(so[i] + sc[i] != so[i + 1] for i in range(len(so) - 1)) or sum(sc) != plane_bytes:
            raise ValueError(f"{path}: page data is not contiguous")
        offsets.append(so[0])
    if any(offsets[i] + plane_bytes != offsets[i + 1] for i in range(len(offsets) - 1)):
        raise ValueError(f"{path}: pages are not stored back to back; cannot be memory-mapped")
    desc = ""
    if 270 in p0:
        desc = p0[270][2].split(b"\0")[0].decode("utf-8", "replace")
    n = len(pages)
    dim_res = {"X": None, "Y": None, "Z": None, "T": None}
    shape, axes, text = (n, y, x), "ZYX", ""
    if "<OME" in desc:
        def attr(name, cast, default=None):
            m = re.search(rf'\b{name}="([^"]*)"', desc)
            return cast(m.group(1)) if m else default
        sz, st = attr("SizeZ", int, 1), attr("SizeT", int, 1)
        order = attr("DimensionOrder", str, "XYZTC")
        if sz * st != n:
            raise ValueError(f"{path}: OME sizes do not match the page count")
        zt = [c for c in order[2:] if c in "ZT"]
        shape = (st, sz, y, x) if zt == ["Z", "T"] else (sz, st, y, x)
        axes = "TZYX" if zt == ["Z", "T"] else "ZTYX"
        for ax in ("X", "Y", "Z"):
            dim_res[ax] = attr(f"PhysicalSize{ax}", float)
        dim_res["T"] = attr("TimeIncrement", float)
        m = re.search(r"<Description>(.*?)</Description>", desc, re.S)
        text = m.group(1) if m else ""
    elif n == 1:
        shape, axes = (y, x), "YX"
    return TiffLayout(offsets[0], dtype, shape, axes, dim_res, text)


def memmap(path, mode="r+"):
    """numpy.memmap over the sample block (what tifffile.memmap returns for such a file), plus the layout."""
    lay = read_layout(path)
    return np.memmap(path, dtype=lay.dtype, mode=mode, offset=lay.offset, shape=lay.shape), lay
