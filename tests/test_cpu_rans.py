"""Host-side entropy coder against golden vectors produced by the reference's own pybind11 extensions
and EntropyBottleneck (tests/golden/make_rans_goldens.py).  No GPU needed: the coder is host code of libmasic_hip.so."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rans_vectors.npz"))


def _rans():
    from masic_amd import rans
    return rans


def test_pmf_to_quantized_cdf_matches_reference():
    rans = _rans()
    n = 0
    for key in G.files:
        if not key.startswith("pmf"):
            continue
        prec = int(key[3:5])
        cdf = rans.pmf_to_quantized_cdf(G[key], prec)
        ref = G["cdf" + key[3:]]
        assert cdf.tolist() == ref.tolist(), key
        assert cdf[0] == 0 and cdf[-1] == 1 << prec and np.all(np.diff(cdf.astype(np.int64)) > 0)
        n += 1
    assert n >= 10
    with pytest.raises(RuntimeError):
        rans.pmf_to_quantized_cdf([0.5, float("nan")], 16)
    with pytest.raises(RuntimeError):
        rans.pmf_to_quantized_cdf([0.0, 0.0], 16)


@pytest.mark.parametrize("case", ["short", "mid", "long", "escapes"])
def test_rans_streams_byte_exact_and_round_trip(case):
    rans = _rans()
    tab, sizes, offs = G["tables"], G["sizes"], G["offsets"]
    sym, idx, ref = G["sym_" + case], G["idx_" + case], G["enc_" + case].tobytes()
    enc = rans.encode_with_indexes(sym, idx, tab, sizes, offs)
    assert enc == ref, f"{case}: stream differs from the reference's ({len(enc)} vs {len(ref)} bytes)"
    assert rans.decode_with_indexes(ref, idx, tab, sizes, offs).tolist() == sym.tolist()
    # ragged list-of-lists tables (the reference's calling convention) give the same stream
    rows = [tab[i, :sizes[i]].tolist() for i in range(len(sizes))]
    assert rans.encode_with_indexes(sym.tolist(), idx.tolist(), rows, sizes.tolist(), offs.tolist()) == ref


def test_rans_edge_cases_and_errors():
    rans = _rans()
    tab, sizes, offs = G["tables"], G["sizes"], G["offsets"]
    empty = rans.encode_with_indexes([], [], tab, sizes, offs)
    assert len(empty) == 8 and rans.decode_with_indexes(empty, [], tab, sizes, offs).size == 0      # just the flushed state
    extreme = np.array([2 ** 30, -(2 ** 30), 0], dtype=np.int32)
    idx = np.array([0, 1, 2], dtype=np.int32)
    assert rans.decode_with_indexes(rans.encode_with_indexes(extreme, idx, tab, sizes, offs), idx, tab, sizes, offs).tolist() == extreme.tolist()
    with pytest.raises(RuntimeError):
        rans.encode_with_indexes([1], [len(sizes)], tab, sizes, offs)                               # table index out of range
    bad = tab.copy(); bad[0, 1] = bad[0, 0]
    with pytest.raises(RuntimeError):
        rans.encode_with_indexes([1], [0], bad, sizes, offs)                                        # not strictly increasing
    enc = rans.encode_with_indexes(G["sym_mid"], G["idx_mid"], tab, sizes, offs)
    with pytest.raises(RuntimeError):
        rans.decode_with_indexes(enc[:16], G["idx_mid"], tab, sizes, offs)                          # truncated stream


def test_compressai_ans_and_cxx_shims():
    from compressai import _CXX, ans
    tab, sizes, offs = G["tables"], G["sizes"], G["offsets"]
    rows = [tab[i, :sizes[i]].tolist() for i in range(len(sizes))]
    sym, idx = G["sym_short"].tolist(), G["idx_short"].tolist()
    enc = ans.RansEncoder().encode_with_indexes(sym, idx, rows, sizes.tolist(), offs.tolist())
    assert enc == G["enc_short"].tobytes()
    assert ans.RansDecoder().decode_with_indexes(enc, idx, rows, sizes.tolist(), offs.tolist()) == sym
    b = ans.BufferedRansEncoder()
    b.encode_with_indexes(sym[:3], idx[:3], rows, sizes.tolist(), offs.tolist())
    b.encode_with_indexes(sym[3:], idx[3:], rows, sizes.tolist(), offs.tolist())
    assert b.flush() == enc
    assert _CXX.pmf_to_quantized_cdf(G["pmf16_1"].tolist(), 16) == G["cdf16_1"].tolist()


def test_entropy_bottleneck_update_tables_match_reference():
    """EntropyBottleneck.update(): offsets, lengths and quantised CDFs of the reference for the same parameters (host arithmetic)."""
    import torch
    from compressai.entropy_models import EntropyBottleneck
    eb = EntropyBottleneck(12)
    sd = {k[len("eb_state/"):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("eb_state/")}
    for k in ("_offset", "_quantized_cdf", "_cdf_length"):
        ref = sd.pop(k)
        eb_ref = ref
        setattr(eb, "_ref" + k, eb_ref)
    eb.load_state_dict(sd, strict=False)
    eb.update(force=True)
    assert torch.equal(eb._offset.cpu(), eb._ref_offset)
    assert torch.equal(eb._cdf_length.cpu(), eb._ref_cdf_length)
    assert torch.equal(eb._quantized_cdf.cpu(), eb._ref_quantized_cdf)
    eb.update()                                           # second call is a no-op (reference :305-306)
    assert torch.equal(eb._quantized_cdf.cpu(), eb._ref_quantized_cdf)


def test_adaptive_rans_round_trip_in_wavefront_chunks():
    """masic_rans_encode_freqs / masic_rans_decoder_*: one table per symbol (the y streams of HSIC.compress), decoded
    incrementally in chunks of rows as the wavefront decoder does; corrupt input is reported, not read past."""
    from masic_amd import codec
    rs = np.random.RandomState(0)
    n, L = 5000, 41
    cuts = np.sort(rs.choice(np.arange(1, 65536), size=(n, L - 1), replace=True), axis=1)
    for r in range(n):                                    # strictly increasing starts
        while len(np.unique(cuts[r])) < L - 1:
            cuts[r] = np.sort(rs.choice(np.arange(1, 65536), size=L - 1, replace=False))
    starts = np.concatenate([np.zeros((n, 1), dtype=np.int64), cuts], axis=1)
    ends = np.concatenate([starts[:, 1:], np.full((n, 1), 65536)], axis=1)
    p = (ends - starts) / 65536.0
    sym = np.array([rs.choice(L, p=p[r]) for r in range(n)])
    sf = np.stack([starts[np.arange(n), sym], (ends - starts)[np.arange(n), sym]], axis=1).astype(np.int32)
    data = codec.encode_freqs(sf)
    ideal = float(-np.log2(p[np.arange(n), sym]).sum()) / 8
    assert len(data) <= ideal * 1.01 + 16                 # rANS: within a percent of the ideal code length
    dec = codec.AdaptiveDecoder(data)
    got, r = [], 0
    for chunk in (1, 7, 300, 1000, n):
        hi = min(n, r + chunk)
        got.append(dec.decode_rows(starts[r:hi].astype(np.uint16)))
        r = hi
    dec.close()
    assert (np.concatenate(got) == sym).all()
    with pytest.raises(RuntimeError):
        codec.encode_freqs(np.array([[0, 0]], dtype=np.int32))             # empty interval
    dec = codec.AdaptiveDecoder(data[:8])
    with pytest.raises(RuntimeError):
        for _ in range(64):                                               # the truncated stream runs dry
            dec.decode_rows(starts[:100].astype(np.uint16))
    dec.close()


def test_wavefront_order_is_causal_for_the_type_a_mask():
    """every latent pixel exactly once, and the 12 causal neighbours of the 5x5 type-A mask in strictly earlier steps"""
    from masic_amd import codec
    for h, w in ((1, 1), (4, 4), (8, 12), (5, 3), (32, 32)):
        steps = codec.wavefront_steps(h, w)
        when = np.full(h * w, -1)
        for t, pix in enumerate(steps):
            assert (when[pix] == -1).all() and (np.diff(pix) > 0).all()
            when[pix] = t
        assert (when >= 0).all() and len(steps) == w + 3 * (h - 1)
        for i in range(h):
            for j in range(w):
                for di in (-2, -1, 0):
                    for dj in range(-2, 3):
                        if di == 0 and dj >= 0:
                            continue
                        a, b = i + di, j + dj
                        if 0 <= a < h and 0 <= b < w:
                            assert when[a * w + b] < when[i * w + j]


def test_streaming_decoder_matches_reference_streams():
    """compressai.ans.RansDecoder.set_stream / decode_stream (reference rans_interface.cpp:286-353): the reference's own streams decoded in
    uneven pieces, with the coder state kept between calls, give the symbols of the one-shot decode."""
    from compressai.ans import RansDecoder
    tab, sizes, offs = G["tables"], G["sizes"], G["offsets"]
    for case in ("short", "long", "escapes"):
        sym, idx, ref = G["sym_" + case], G["idx_" + case], G["enc_" + case].tobytes()
        dec = RansDecoder()
        with pytest.raises(RuntimeError):
            RansDecoder().decode_stream(idx[:1].tolist(), tab, sizes, offs)
        dec.set_stream(ref)
        got, pos = [], 0
        for n in (1, 3, 0, 17, 10 ** 9):
            piece = idx[pos:pos + n]
            got += dec.decode_stream(piece.tolist(), tab, sizes, offs)
            pos += len(piece)
        assert got == sym.tolist(), case


def test_gaussian_conditional_tables_vs_reference():
    """GaussianConditional.update / update_scale_table (reference entropy_models.py:494-525): the quantised CDF tables of a scale table
    against the reference's (tests/golden/misc_api.npz); host code, no GPU needed."""
    from compressai.entropy_models import GaussianConditional
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "misc_api.npz"))
    gc = GaussianConditional([float(v) for v in fx["gc/scale_table"]])
    gc.update()
    assert gc._quantized_cdf.numpy().tolist() == fx["gc/quantized_cdf"].tolist()
    assert gc._offset.numpy().tolist() == fx["gc/offset"].tolist() and gc._cdf_length.numpy().tolist() == fx["gc/cdf_length"].tolist()
    gc2 = GaussianConditional(None)
    gc2.update_scale_table([float(v) for v in fx["gc/scale_table"]])
    assert gc2._quantized_cdf.numpy().tolist() == fx["gc/quantized_cdf"].tolist()
    gc2.update_scale_table([1.0, 2.0])                       # already initialised and not forced: unchanged (reference :497-499)
    assert gc2._quantized_cdf.shape == gc._quantized_cdf.shape


def test_per_channel_streams_decode_with_the_host_decoder():
    """masic_rans_encode_channels (container MSR2: one rANS stream per latent channel, what the device decoder reads): every channel's
    stream, decoded with the host decoder over that channel's tables, gives back the channel's symbols; the framing splits exactly."""
    from masic_amd import codec
    rs = np.random.RandomState(11)
    npix, nch, L = 97, 7, 21
    starts = np.zeros((npix * nch, L), dtype=np.int64)
    sym = np.zeros(npix * nch, dtype=np.int64)
    sf = np.zeros((npix * nch, 2), dtype=np.int32)
    for r in range(npix * nch):
        f = rs.randint(1, 200, size=L).astype(np.int64)
        f[rs.randint(L)] += 65536 - f.sum()
        assert f.min() >= 1 and f.sum() == 65536
        starts[r] = np.concatenate([[0], np.cumsum(f)[:-1]])
        sym[r] = rs.randint(L)
        sf[r] = (starts[r, sym[r]], f[sym[r]])
    blob = codec.encode_channels(sf, npix, nch) + b"tail"
    streams, used = codec.split_channels(blob)
    assert used == len(blob) - 4 and len(streams) == nch and all(len(s) >= 8 and len(s) % 4 == 0 for s in streams)
    for c, data in enumerate(streams):
        rows = np.arange(npix) * nch + c
        dec = codec.AdaptiveDecoder(data)
        got = dec.decode_rows(starts[rows].astype(np.uint16))
        dec.close()
        assert np.array_equal(got, sym[rows]), c
    empty, used = codec.split_channels(codec.encode_channels(np.zeros((0, 2), dtype=np.int32), 0, 0))
    assert empty == [] and used == 4
