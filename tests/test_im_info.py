"""CPU tests of the on-disk layer: BigTIFF structure, round trips, Nellie's naming and canonical axes."""
import os
import struct

import numpy as np
import pytest

from nellie_amd.im_info import ome_tiff
from nellie_amd.im_info.verifier import ImInfo


def test_bigtiff_structure_and_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    data = rng.integers(0, 60000, (3, 4, 10, 12)).astype(np.uint16)
    path = str(tmp_path / "a.ome.tif")
    off = ome_tiff.create(path, data.shape, np.uint16, {"X": 0.1, "Y": 0.1, "Z": 0.3, "T": 2.0}, "hello <&>", data=data)
    raw = open(path, "rb").read()
    assert raw[:4] == b"II" + struct.pack("<H", 43) and struct.unpack("<HH", raw[4:8]) == (8, 0)
    first_ifd, = struct.unpack("<Q", raw[8:16])
    assert off == 16 and first_ifd == 16 + data.nbytes
    n_entries, = struct.unpack("<Q", raw[first_ifd:first_ifd + 8])
    tags = [struct.unpack("<H", raw[first_ifd + 8 + 20 * k:first_ifd + 10 + 20 * k])[0] for k in range(n_entries)]
    assert tags == sorted(tags) and {256, 257, 258, 259, 262, 270, 273, 277, 278, 279, 339} <= set(tags)
    mm, lay = ome_tiff.memmap(path, mode="r")
    assert lay.axes == "TZYX" and mm.shape == data.shape and mm.dtype == np.uint16
    assert np.array_equal(mm, data)
    assert lay.dim_res == {"X": 0.1, "Y": 0.1, "Z": 0.3, "T": 2.0}
    assert lay.description == "hello &lt;&amp;&gt;"
    # planes back to back, one strip each
    assert len(ome_tiff._read_ifds(open(path, "rb"), True, "<")) == 12


def test_zero_filled_allocation_is_writable(tmp_path):
    path = str(tmp_path / "z.ome.tif")
    ome_tiff.create(path, (2, 3, 8, 8), np.float32, {"X": 0.2, "Y": 0.2, "Z": 0.2, "T": 1.0}, "frangi filtered im")
    mm, lay = ome_tiff.memmap(path, mode="r+")
    assert mm.dtype == np.float32 and not mm.any()
    mm[1, 2] = 7.5
    mm.flush()
    del mm
    again, _ = ome_tiff.memmap(path, mode="r")
    assert again[1, 2, 3, 3] == 7.5 and again[0].sum() == 0


def test_create_removes_the_temporaries_of_writers_that_died(tmp_path):
    """create() writes under `<path>.tmp<pid>_<hex>` and moves the file into place; a process killed in between leaves the
    temporary behind.  The next create() of the same path removes those whose writer is gone, and nothing else."""
    import subprocess
    import sys
    path = str(tmp_path / "o.ome.tif")
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    here = ome_tiff._host_tag()
    stale = f"{path}.tmp{dead.pid}_{here}_0123abcd"
    mine = f"{path}.tmp{os.getpid()}_{here}_89abcdef"          # a writer that is alive (this process) and recent: kept
    other = f"{path}.tmpnotes"
    # the same dead pid in a temporary ANOTHER host wrote (shared file system) says nothing here -- kept while it is recent,
    # removed once it is an hour old; the names of round 5 (no host tag) are treated the same way
    elsewhere, elsewhere_old, legacy = f"{path}.tmp{dead.pid}_deadbeef_0123abcd", f"{path}.tmp{dead.pid}_deadbeef_76543210", f"{path}.tmp{dead.pid}_0123abcd"
    for p in (stale, mine, other, elsewhere, elsewhere_old, legacy):
        open(p, "wb").write(b"x")
    os.utime(elsewhere_old, (1.0e9, 1.0e9))
    ome_tiff.create(path, (1, 2, 4, 4), np.uint8, {"X": 1.0, "Y": 1.0, "Z": 1.0, "T": 1.0}, "d")
    assert not os.path.exists(stale) and os.path.exists(mine) and os.path.exists(other) and os.path.exists(path)
    assert os.path.exists(elsewhere) and os.path.exists(legacy) and not os.path.exists(elsewhere_old)


def test_iminfo_layout_matches_nellie_convention(tmp_path):
    vol = np.arange(2 * 3 * 4 * 5, dtype=np.float32).reshape(2, 3, 4, 5)
    im = ImInfo(vol, dim_res={"X": 0.1, "Y": 0.1, "Z": 0.25, "T": 1.5}, output_dir=str(tmp_path), name="cell")
    assert im.axes == "TZYX" and im.shape == (2, 3, 4, 5) and not im.no_z and not im.no_t
    nn = os.path.join(str(tmp_path), "nellie_output", "nellie_necessities")
    assert im.im_path == os.path.join(nn, "cell-TZYX-T1p5_Z0p25_Y0p1_X0p1-ch0-t0_to_1.ome.tif")
    assert im.pipeline_paths["im_preprocessed"].endswith("-ch0-t0_to_1-im_preprocessed.ome.tif")
    assert im.pipeline_paths["features_organelles"].startswith(os.path.join(str(tmp_path), "nellie_output", "cell-"))
    assert np.array_equal(im.get_memmap(im.im_path), vol)
    out = im.allocate_memory(im.pipeline_paths["im_instance_label"], dtype="int32", description="instance segmentation",
                             return_memmap=True)
    assert out.shape == vol.shape and out.dtype == np.int32
    out[1] = 3
    out.flush()
    assert im.get_memmap(im.pipeline_paths["im_instance_label"])[1].min() == 3
    # a 3-D source gains a leading T axis (verifier.py:889-929); reopening the saved TIFF gives the same view
    im3 = ImInfo(vol[0], dim_res={"X": 0.1, "Y": 0.1, "Z": 0.25, "T": None}, output_dir=str(tmp_path), name="single")
    assert im3.axes == "TZYX" and im3.shape == (1, 3, 4, 5) and im3.no_t and "-t" not in os.path.basename(im3.im_path)
    again = ImInfo(im3.im_path, output_dir=str(tmp_path / "again"))
    assert again.shape == (1, 3, 4, 5) and again.dim_res["Z"] == 0.25
    im2 = ImInfo(vol[0, 0], dim_res={"X": 0.1, "Y": 0.1}, output_dir=str(tmp_path), name="flat")
    assert im2.axes == "TYX" and im2.no_z
    im.remove_intermediates()
    assert not os.path.exists(im.im_path)


def test_rejects_what_cannot_be_mapped(tmp_path):
    p = tmp_path / "bad.tif"
    p.write_bytes(b"II" + struct.pack("<HI", 42, 8) + struct.pack("<H", 0) + struct.pack("<I", 0))
    with pytest.raises(ValueError):
        ome_tiff.read_layout(str(p))
    with pytest.raises(ValueError):
        ome_tiff.create(str(tmp_path / "c.ome.tif"), (1, 1, 2, 2), np.complex64)


def test_files_read_back_with_an_independent_tiff_reader(tmp_path):
    """Pillow's TIFF reader (libtiff-style parser, shares no code with ours) sees every plane, the pixel type, the
    pixel data and the OME-XML ImageDescription of the BigTIFF files this package writes."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageSequence
    import xml.etree.ElementTree as ET
    from nellie_amd.im_info import ome_tiff
    rng = np.random.default_rng(0)
    for dtype, mode in ((np.float32, "F"), (np.int32, "I"), (np.uint16, "I;16")):
        a = (rng.random((2, 3, 20, 24)) * 1000).astype(dtype)
        path = str(tmp_path / f"t_{np.dtype(dtype).name}.ome.tif")
        ome_tiff.create(path, a.shape, dtype, dim_res={"X": 0.1, "Y": 0.2, "Z": 0.3, "T": 1.5}, description="d", data=a)
        with Image.open(path) as im:
            assert im.format == "TIFF" and im.n_frames == 6 and im.size == (24, 20) and im.mode == mode
            got = np.stack([np.array(fr) for fr in ImageSequence.Iterator(im)]).reshape(a.shape)
            assert got.dtype == np.dtype(dtype) and np.array_equal(got, a)
            im.seek(0)
            xml = im.tag_v2[270]
        root = ET.fromstring(xml)                                  # well-formed
        px = [e for e in root.iter() if e.tag.endswith("Pixels")][0]
        assert (px.get("SizeX"), px.get("SizeY"), px.get("SizeZ"), px.get("SizeT")) == ("24", "20", "3", "2")
        assert float(px.get("PhysicalSizeX")) == 0.1 and float(px.get("PhysicalSizeY")) == 0.2
        assert float(px.get("PhysicalSizeZ")) == 0.3 and float(px.get("TimeIncrement")) == 1.5


@pytest.mark.parametrize("dtype,ome", [(np.float32, "float"), (np.int32, "int32"), (np.uint16, "uint16"), (np.uint8, "uint8"), (np.float64, "double")])
@pytest.mark.parametrize("shape", [(1, 1, 9, 11), (1, 5, 8, 8), (3, 1, 6, 10), (2, 4, 7, 9)])
def test_written_files_meet_tifffile_memmap_requirements(tmp_path, dtype, ome, shape):
    """The writer's files pass an independent strict checker of what `tifffile.memmap` and `ome_types.from_xml` need
    (tests/tiff_conformance.py), and the bytes at the reported data offset are the array."""
    from tiff_conformance import check_memmappable_ome_bigtiff
    rng = np.random.default_rng(1)
    data = (rng.random(shape) * 100).astype(dtype)
    path = str(tmp_path / "w.ome.tif")
    dr = {"X": 0.1, "Y": 0.2, "Z": 0.3 if shape[1] > 1 else None, "T": 1.5 if shape[0] > 1 else None}
    ome_tiff.create(path, shape, dtype, dr, "frangi filtered im", data=data)
    info = check_memmappable_ome_bigtiff(path)
    assert info["dtype"] == np.dtype(dtype) and info["shape"] == shape and info["pixel_type"] == ome
    assert info["description"] == "frangi filtered im"
    assert info["dim_res"]["X"] == 0.1 and info["dim_res"]["Y"] == 0.2
    assert ("Z" in info["dim_res"]) == (dr["Z"] is not None) and ("T" in info["dim_res"]) == (dr["T"] is not None)
    mapped = np.memmap(path, dtype=info["dtype"], mode="r", offset=info["offset"], shape=info["shape"])     # what tifffile.memmap returns
    assert np.array_equal(mapped, data)


def test_stage_outputs_meet_the_requirements(tmp_path):
    from tiff_conformance import check_memmappable_ome_bigtiff
    vol = np.arange(2 * 3 * 4 * 5, dtype=np.uint16).reshape(2, 3, 4, 5)
    im = ImInfo(vol, dim_res={"X": 0.1, "Y": 0.1, "Z": 0.25, "T": 1.5}, output_dir=str(tmp_path), name="cell")
    assert check_memmappable_ome_bigtiff(im.im_path)["shape"] == (2, 3, 4, 5)
    for stage, dt in (("im_preprocessed", "float32"), ("im_instance_label", "int32")):
        out = im.allocate_memory(im.pipeline_paths[stage], dtype=dt, description=stage, return_memmap=True)
        out[...] = 3
        out.flush()
        info = check_memmappable_ome_bigtiff(im.pipeline_paths[stage])
        assert info["dtype"] == np.dtype(dt) and info["shape"] == (2, 3, 4, 5) and info["description"] == stage


def test_array_sources_never_reuse_a_stale_canonical_copy(tmp_path):
    """Two different arrays with the same shape / axes / resolutions in the same directory: each ImInfo sees its own pixels."""
    a = np.full((2, 3, 4), 5, np.float32)
    b = np.full((2, 3, 4), 9, np.float32)
    ia = ImInfo(a, dim_res={"X": 1, "Y": 1, "Z": 1, "T": None}, output_dir=str(tmp_path))
    assert float(np.asarray(ia.im).max()) == 5.0
    ib = ImInfo(b, dim_res={"X": 1, "Y": 1, "Z": 1, "T": None}, output_dir=str(tmp_path))
    assert ia.im_path == ib.im_path and float(np.asarray(ib.im).min()) == 9.0


def test_detailed_names_equal_the_reference_strings():
    """tests/golden/naming_cases.json: strings produced by the reference's FileInfo._get_output_path /
    ImInfo.create_output_path (verifier.py:574-618, 805-828) on the same inputs."""
    import json
    from nellie_amd.im_info.verifier import detailed_output_name
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "naming_cases.json")))
    from nellie_amd.im_info.verifier import FileInfo
    assert len(cases) >= 11
    for c in cases:
        if c.get("filename"):          # filename_no_ext as the reference's FileInfo.__init__ derives it ("x.ome.tif" -> "x.ome")
            assert FileInfo(os.path.join("somewhere", c["filename"]), output_naming=c["naming"]).filename_no_ext == c["name"]
        name = c["name"] if c.get("naming") == "stable" else \
            detailed_output_name(c["name"], c["axes"], c["dim_res"], c["ch"], c["t_start"], c["t_end"])
        assert os.path.join("OUT", name) == c["user_no_ext"]
        assert os.path.join("OUT", "nellie_necessities", name) == c["necessities_no_ext"]
        assert os.path.join("OUT", "nellie_necessities", name) + ".ome.tif" == c["ome_output_path"]
        for stage, ref in c["pipeline_paths"].items():
            base = c["user_no_ext"] if stage == "features_organelles" else c["necessities_no_ext"]
            assert f"{base}-{stage}{'.csv' if stage == 'features_organelles' else '.ome.tif'}" == ref


def test_run_signature_takes_a_file_info(tmp_path):
    """nellie.run.run(file_info, ...) (run.py:18-26, 49): FileInfo -> ImInfo, channel / time selection, detailed names."""
    from nellie_amd.im_info.verifier import FileInfo
    vol = (np.arange(4 * 3 * 6 * 7) % 251).astype(np.uint16).reshape(4, 3, 6, 7)
    src = str(tmp_path / "stack.ome.tif")
    ome_tiff.create(src, vol.shape, np.uint16, {"X": 0.2, "Y": 0.2, "Z": 0.5, "T": 2.0}, "raw", data=vol)
    fi = FileInfo(src, output_dir=str(tmp_path / "out"))
    fi.find_metadata(); fi.load_metadata()
    assert fi.axes == "TZYX" and fi.shape == vol.shape and fi.dim_res["Z"] == 0.5 and fi.good_dims and fi.good_axes
    fi.select_temporal_range(1, 2)
    im = ImInfo(fi)
    assert im.shape == (2, 3, 6, 7) and np.array_equal(np.asarray(im.im), vol[1:3])
    assert os.path.basename(im.im_path) == "stack.ome-TZYX-T2p0_Z0p5_Y0p2_X0p2-ch0-t1_to_2.ome.tif"
    assert im.im_path.startswith(os.path.join(str(tmp_path / "out"), "nellie_output", "nellie_necessities"))
    # the canonical copy of a selection is reused (its name carries the range), not rewritten
    mtime = os.path.getmtime(im.im_path)
    im_again = ImInfo(fi)
    assert im_again.im_path == im.im_path and os.path.getmtime(im.im_path) == mtime
    # an out-of-range selection is an error, as in the reference (verifier.py:475-506, 393-408)
    with pytest.raises(IndexError):
        fi.select_temporal_range(0, 9)
    with pytest.raises(ValueError):
        fi.select_temporal_range(3, 1)
    fi.t_start, fi.t_end = 0, 9
    with pytest.raises(ValueError):
        ImInfo(fi)
    # "stable" naming = the bare file name (verifier.py:597-598)
    fs = FileInfo(src, output_dir=str(tmp_path / "out_stable"), output_naming="stable")
    ims = ImInfo(fs)
    assert os.path.basename(ims.im_path) == "stack.ome.ome.tif"
    assert os.path.basename(ims.pipeline_paths["im_instance_label"]) == "stack.ome-im_instance_label.ome.tif"
    import inspect
    from nellie_amd.run import run
    assert list(inspect.signature(run).parameters)[:7] == ["file_info", "remove_edges", "otsu_thresh_intensity", "threshold", "timeit", "device", "low_memory"]
