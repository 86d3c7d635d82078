"""Packed outputs (include/nellie_amd.h: nl_outputs_pack / nl_outputs_unpack).  The host half -- the expansion of a blob into
dense arrays -- needs no GPU: a numpy model of the packer builds blobs for adversarial rows; the device half is checked
against it and against the dense download in the gpu test below."""
import numpy as np
import pytest

MAGIC = 0x4b43415031304c4e


def pad16(b):
    return (b + 15) & ~15


def pack_model(frangi, labels=None):
    """The layout nl_outputs_pack produces, built with numpy."""
    nz, ny, nx = frangi.shape
    rows, wpr = nz * ny, (nx + 63) // 64

    def bits_of(mask):
        padded = np.zeros((rows, wpr * 64), bool)
        padded[:, :nx] = mask.reshape(rows, nx)
        return np.packbits(padded.reshape(rows, wpr, 8, 8)[:, :, ::-1, ::-1].reshape(rows, wpr, 64), axis=-1, bitorder="big").view(">u8").astype("<u8").reshape(rows, wpr)

    f = frangi.reshape(rows, nx)
    fmask = f.view(np.uint32) != 0
    fb = bits_of(fmask)
    fo = np.concatenate([[0], np.cumsum(fmask.sum(1))]).astype(np.uint32)
    fv = f[fmask].astype(np.float32)
    parts = {"fb": fb.tobytes(), "fo": fo.tobytes(), "fv": fv.tobytes(), "lb": b"", "lo": b"", "lr": b""}
    n_runs = 0
    if labels is not None:
        l = labels.reshape(rows, nx)
        lmask = l > 0
        starts = lmask & ~np.concatenate([np.zeros((rows, 1), bool), lmask[:, :-1]], axis=1)
        parts["lb"] = bits_of(lmask).tobytes()
        parts["lo"] = np.concatenate([[0], np.cumsum(starts.sum(1))]).astype(np.uint32).tobytes()
        parts["lr"] = l[starts].astype(np.int32).tobytes()
        n_runs = int(starts.sum())
    off, order, offs = pad16(128), ("fb", "lb", "fo", "lo", "fv", "lr"), {}
    for k in order:
        offs[k] = off
        off += pad16(len(parts[k]))
    hdr = np.zeros(16, np.int64)
    hdr[:8] = [MAGIC, nz, ny, nx, wpr, fv.size, n_runs, 0 if labels is None else 1]
    hdr[8:15] = [offs["fb"], offs["lb"], offs["fo"], offs["lo"], offs["fv"], offs["lr"], off]
    blob = bytearray(off)
    blob[:128] = hdr.tobytes()
    for k in order:
        blob[offs[k]:offs[k] + len(parts[k])] = parts[k]
    return np.frombuffer(bytes(blob), np.uint8).copy()


def volumes(shape, seed, density=0.03):
    rng = np.random.default_rng(seed)
    nz, ny, nx = shape
    fr = np.where(rng.random(shape) < density, rng.random(shape, dtype=np.float32) + np.float32(1e-6), np.float32(0)).astype(np.float32)
    lab = np.zeros(shape, np.int32)
    for _ in range(max(4, int(density * nz * ny * 2))):          # X-runs, some crossing 64-voxel words, some ending at the row end
        z, y = rng.integers(nz), rng.integers(ny)
        x0 = int(rng.integers(nx)); ln = int(rng.integers(1, 150))
        if (lab[z, y, max(x0 - 1, 0):x0 + ln + 1] == 0).all():
            lab[z, y, x0:x0 + ln] = rng.integers(1, 1 << 20)
    lab[0, 0, :] = 7                                              # a run that is a whole row
    if nz * ny > 1:                                               # a run of one voxel at the very end of a row
        lab[-1, -1, max(nx - 2, 0):] = 0
        lab[-1, -1, nx - 1] = 9
    fr[0, 0, 0] = np.float32(-0.0) if nx > 1 else fr[0, 0, 0]   # a non-zero bit pattern that compares equal to zero
    return fr, lab


@pytest.mark.parametrize("shape", [(3, 5, 64), (2, 7, 1), (4, 6, 200), (2, 3, 4100), (5, 9, 63), (1, 1, 129)])
@pytest.mark.parametrize("zero_fill", [True, False])
def test_unpack_expands_model_blobs(shape, zero_fill):
    from nellie_amd import hipnative
    fr, lab = volumes(shape, sum(shape))
    blob = pack_model(fr, lab)
    out_f = np.zeros(shape, np.float32) if not zero_fill else np.full(shape, 5.0, np.float32)
    out_l = np.zeros(shape, np.int32) if not zero_fill else np.full(shape, -3, np.int32)
    hipnative.outputs_unpack(blob, blob.nbytes, out_f, out_l, zero_fill=zero_fill, threads=3)
    assert np.array_equal(out_f.view(np.uint32), fr.view(np.uint32))      # bit patterns: -0.0 survives
    assert np.array_equal(out_l, lab)
    only_f = np.full(shape, 1.0, np.float32)
    hipnative.outputs_unpack(blob, blob.nbytes, only_f, None, zero_fill=True, threads=1)
    assert np.array_equal(only_f.view(np.uint32), fr.view(np.uint32))


def test_unpack_rejects_what_is_not_a_blob():
    from nellie_amd import hipnative
    out = np.zeros((2, 2, 2), np.float32)
    with pytest.raises(ValueError):
        hipnative.outputs_unpack(np.zeros(256, np.uint8), 256, out)
    fr, lab = volumes((2, 2, 2), 1)
    blob = pack_model(fr, None)
    with pytest.raises(ValueError):                                      # no labels inside
        hipnative.outputs_unpack(blob, blob.nbytes, out, np.zeros((2, 2, 2), np.int32))
    with pytest.raises(ValueError):                                      # truncated
        hipnative.outputs_unpack(blob, blob.nbytes - 16, out)


def test_unpack_never_trusts_the_header_for_the_destination():
    """A blob from a context of another shape, or with damaged section offsets / counts, is refused before a single element
    of the caller's arrays is written."""
    from nellie_amd import hipnative
    fr, lab = volumes((3, 4, 70), 2)
    blob = pack_model(fr, lab)
    small = np.full((2, 4, 70), 7.0, np.float32)
    with pytest.raises(ValueError, match="destination"):
        hipnative.outputs_unpack(blob, blob.nbytes, small)
    assert (small == 7.0).all()
    with pytest.raises(AssertionError):                                  # labels of another shape than the frame
        hipnative.outputs_unpack(blob, blob.nbytes, np.zeros((3, 4, 70), np.float32), np.zeros((3, 4, 71), np.int32))
    hdr_fields = {"off_fb": 8, "off_lb": 9, "off_fo": 10, "off_lo": 11, "off_fv": 12, "off_lr": 13}
    for name, k in hdr_fields.items():
        bad = blob.copy()
        bad[:128].view(np.int64)[k] = blob.nbytes - 8                    # the section would run past the end
        out = np.full((3, 4, 70), 7.0, np.float32)
        with pytest.raises(ValueError):
            hipnative.outputs_unpack(bad, bad.nbytes, out, np.zeros((3, 4, 70), np.int32))
        assert (out == 7.0).all(), name
    for k in (5, 6):                                                     # n_values / n_runs smaller than the row offsets say
        bad = blob.copy()
        bad[:128].view(np.int64)[k] = 1
        with pytest.raises(ValueError):
            hipnative.outputs_unpack(bad, bad.nbytes, np.zeros((3, 4, 70), np.float32), np.zeros((3, 4, 70), np.int32))
    bad = blob.copy()
    bad[:128].view(np.int64)[5] = 1 << 40                                # ... or absurdly large
    with pytest.raises(ValueError):
        hipnative.outputs_unpack(bad, bad.nbytes, np.zeros((3, 4, 70), np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(24, 48, 48), (9, 33, 200), (3, 4, 4100), (40, 64, 129)])
def test_device_pack_equals_dense_download(hip, shape):
    """nl_outputs_pack on real products of the path: the blob equals the numpy model's byte for byte, and its expansion
    equals the dense downloads."""
    from nellie_amd import hipnative
    from nellie_amd import pipeline as pl
    from nellie_amd.synthetic import ISO_01, make_volume
    pipe = pl.FramePipeline(shape)
    pipe.filter(make_volume(shape, 11), pl.FilterParams(dim_res=ISO_01))
    n = pipe.label(pipe.frangi_threshold(), pl.min_area_pixels_of(ISO_01))
    fr, lab = pipe.download_frangi(), pipe.download_labels()
    nbytes = pipe.ctx.outputs_pack(True)
    assert nbytes > 0
    land = hipnative.PinnedArray((nbytes,), np.uint8)
    pipe.ctx.outputs_fetch_packed_async(land, nbytes)
    pipe.ctx.outputs_wait()
    model = pack_model(fr, lab)
    assert nbytes == model.nbytes
    hdr = model[:128].view(np.int64)
    assert np.array_equal(land.array[:128].view(np.int64), hdr)
    rows, wpr = shape[0] * shape[1], (shape[2] + 63) // 64
    sizes = {8: rows * wpr * 8, 9: rows * wpr * 8, 10: (rows + 1) * 4, 11: (rows + 1) * 4, 12: int(hdr[5]) * 4, 13: int(hdr[6]) * 4}
    for k, size in sizes.items():                                 # every section (the padding between them is not specified)
        o = int(hdr[k])
        assert np.array_equal(land.array[o:o + size], model[o:o + size]), f"section {k}"
    out_f, out_l = np.full(shape, 2.0, np.float32), np.full(shape, -1, np.int32)
    hipnative.outputs_unpack(land, nbytes, out_f, out_l, zero_fill=True, threads=4)
    assert np.array_equal(out_f, fr) and np.array_equal(out_l, lab) and lab.max() == n
    # the other order of the same work (bench.py's host-to-host figure): the fill by nl_host_zero beforehand, a scatter-only unpack
    out_f[...] = 2.0; out_l[...] = -1
    hipnative.host_zero(out_f, threads=3); hipnative.host_zero(out_l, threads=5)
    assert not out_f.any() and not out_l.any()
    hipnative.outputs_unpack(land, nbytes, out_f, out_l, zero_fill=False, threads=4)
    assert np.array_equal(out_f, fr) and np.array_equal(out_l, lab)
    land.free()
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(40, 96, 130), (64, 128, 136)])
def test_pack_enqueued_under_labels_wait_equals_the_separate_call(hip, shape):
    """nl_outputs_pack_with_label (round 5): nl_label_run enqueues the frame's pack before its own wait and the outputs_pack that follows only
    reads the size.  Same blob, byte for byte, as the separate call on the same frame -- over three frames of one context (the second blob
    may only overwrite the staging buffer once the first has been fetched) and with an outputs_pack(False) in between (Frangi only: redone)."""
    from nellie_amd import hipnative
    from nellie_amd import pipeline as pl
    from nellie_amd.synthetic import ISO_01, make_volume
    p = pl.FilterParams(dim_res=ISO_01)
    ma = pl.min_area_pixels_of(ISO_01)

    def blobs(with_label):
        pipe = pl.FramePipeline(shape)
        pipe.ctx.outputs_pack_with_label(with_label)
        out = []
        for seed in (11, 12, 13):
            pipe.filter(make_volume(shape, seed), p)
            n = pipe.label(pipe.frangi_threshold(), ma)
            if seed == 12:
                nf = pipe.ctx.outputs_pack(False)                # a Frangi-only blob in between: the pending pack is dropped, this one is computed
                assert 0 < nf
            nbytes = pipe.ctx.outputs_pack(True)
            assert nbytes > 0
            land = hipnative.PinnedArray((nbytes,), np.uint8)
            pipe.ctx.outputs_fetch_packed_async(land, nbytes)
            pipe.ctx.outputs_wait()
            fr, lab = pipe.download_frangi(), pipe.download_labels()
            of, ol = np.full(shape, 2.0, np.float32), np.full(shape, -1, np.int32)
            hipnative.outputs_unpack(land, nbytes, of, ol, zero_fill=True, threads=2)
            assert np.array_equal(of, fr) and np.array_equal(ol, lab) and lab.max() == n
            hdr = land.array[:128].view(np.int64).copy()
            sect = [bytes(land.array[int(hdr[k]):int(hdr[k]) + sz]) for k, sz in ((12, int(hdr[5]) * 4), (13, int(hdr[6]) * 4))]
            out.append((nbytes, hdr, sect))
            land.free()
        pipe.close()
        return out

    a, b = blobs(True), blobs(False)
    for (na, ha, sa), (nb, hb, sb) in zip(a, b):
        assert na == nb and np.array_equal(ha, hb) and sa == sb


@pytest.mark.gpu
def test_dense_frame_does_not_pack(hip):
    """More than a quarter of the voxels non-zero: nbytes == 0, the caller takes the dense download."""
    from nellie_amd import pipeline as pl
    shape = (8, 32, 64)
    pipe = pl.FramePipeline(shape)
    pipe.upload_frangi(np.ones(shape, np.float32))
    pipe.label(0.5, 1)
    assert pipe.ctx.outputs_pack(True) == 0
    pipe.close()
