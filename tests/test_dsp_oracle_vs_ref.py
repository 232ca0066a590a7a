"""CPU, build container only: oracle/dsp_oracle.c and oracle/txfm_oracle.c against the reference's `_c` functions.
Grids follow the reference's tests (test/ResidualTest.cc, SpatialFullDistortionTest.cc, BlockErrorTest.cc, SatdTest.cc,
hadamard_test.cc, VarianceTest.cc, quantize_func_test.cc / QuantAsmTest.cc, FwdTxfm2dAsmTest.cc, InvTxfm2dAsmTest.cc)."""
import ctypes as C

import numpy as np
import pyoracle
import pytest

from txfm_cases import TX_H, TX_W, ref_fwd, ref_inv, residual_block, valid_types

P = C.c_void_p


def p(a):
    return a.ctypes.data_as(P)


SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (16, 64), (64, 16), (32, 8), (128, 128)]


@pytest.mark.parametrize("bd", [8, 10])
def test_residual_and_spatial_sse(ref, oracle, bd):
    rng = np.random.default_rng(bd)
    dt, fr, fo = (np.uint8, "svt_residual_kernel8bit_c", "orc_residual8") if bd == 8 else (np.uint16, "svt_residual_kernel16bit_c", "orc_residual16")
    sse_r, sse_o = ("svt_spatial_full_distortion_kernel_c", "orc_spatial_sse8") if bd == 8 else ("svt_full_distortion_kernel16_bits_c", "orc_spatial_sse16")
    getattr(ref, sse_r).restype = C.c_uint64
    getattr(oracle, sse_o).restype = C.c_uint64
    for (w, h) in SIZES:
        for pat in ("random", "max", "min"):
            s1, s2 = w + 3, w + 8
            a = rng.integers(0, 1 << bd, (h, s1)).astype(dt)
            b = rng.integers(0, 1 << bd, (h, s2)).astype(dt)
            if pat == "max":
                a[:] = (1 << bd) - 1; b[:] = 0
            if pat == "min":
                a[:] = 0; b[:] = (1 << bd) - 1
            ra, rb = np.zeros((h, w + 1), np.int16), np.zeros((h, w + 1), np.int16)
            getattr(ref, fr)(p(a), C.c_uint32(s1), p(b), C.c_uint32(s2), p(ra), C.c_uint32(w + 1), C.c_uint32(w), C.c_uint32(h))
            getattr(oracle, fo)(p(a), C.c_uint32(s1), p(b), C.c_uint32(s2), p(rb), C.c_uint32(w + 1), C.c_uint32(w), C.c_uint32(h))
            assert np.array_equal(ra, rb)
            x = getattr(ref, sse_r)(p(a), C.c_uint32(1), C.c_uint32(s1), p(b), C.c_int32(2), C.c_uint32(s2), C.c_uint32(w - 2), C.c_uint32(h))
            y = getattr(oracle, sse_o)(p(a), C.c_uint32(1), C.c_uint32(s1), p(b), C.c_int32(2), C.c_uint32(s2), C.c_uint32(w - 2), C.c_uint32(h))
            assert x == y


def test_coeff_distortion_block_error_satd(ref, oracle):
    rng = np.random.default_rng(3)
    ref.svt_av1_block_error_c.restype = C.c_int64
    oracle.orc_block_error.restype = C.c_int64
    for (w, h) in SIZES[:10]:
        for scale in (100, 1 << 15, 1 << 20):
            c = rng.integers(-scale, scale, (h, w + 2)).astype(np.int32)
            r = rng.integers(-scale, scale, (h, w + 5)).astype(np.int32)
            oa, ob = np.zeros(2, np.uint64), np.zeros(2, np.uint64)
            ref.svt_full_distortion_kernel32_bits_c(p(c), C.c_uint32(w + 2), p(r), C.c_uint32(w + 5), p(oa), C.c_uint32(w), C.c_uint32(h))
            oracle.orc_full_distortion32(p(c), C.c_uint32(w + 2), p(r), C.c_uint32(w + 5), p(ob), C.c_uint32(w), C.c_uint32(h))
            assert np.array_equal(oa, ob)
            ref.svt_full_distortion_kernel_cbf_zero32_bits_c(p(c), C.c_uint32(w + 2), p(oa), C.c_uint32(w), C.c_uint32(h))
            oracle.orc_full_distortion32_cbf_zero(p(c), C.c_uint32(w + 2), p(ob), C.c_uint32(w), C.c_uint32(h))
            assert np.array_equal(oa, ob)
            if scale <= 1 << 15:
                flat_c, flat_r = np.ascontiguousarray(c[:, :w]).ravel(), np.ascontiguousarray(r[:, :w]).ravel()
                sa, sb = C.c_int64(), C.c_int64()
                ea = ref.svt_av1_block_error_c(p(flat_c), p(flat_r), C.c_ssize_t(w * h), C.byref(sa))
                eb = oracle.orc_block_error(p(flat_c), p(flat_r), C.c_ssize_t(w * h), C.byref(sb))
                assert (ea, sa.value) == (eb, sb.value)
                assert ref.svt_aom_satd_c(p(flat_c), w * h) == oracle.orc_satd(p(flat_c), w * h)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_hadamard(ref, oracle, n):
    rng = np.random.default_rng(n)
    for pat in ("random", "max", "min", "alt"):
        src = rng.integers(-255, 256, (n, n + 3)).astype(np.int16)
        if pat == "max": src[:] = 255
        if pat == "min": src[:] = -255
        if pat == "alt": src[:, ::2] = 255; src[:, 1::2] = -255
        a, b = np.zeros(n * n, np.int32), np.zeros(n * n, np.int32)
        getattr(ref, f"svt_aom_hadamard_{n}x{n}_c")(p(src), C.c_ssize_t(n + 3), p(a))
        getattr(oracle, f"orc_hadamard_{n}x{n}")(p(src), C.c_ssize_t(n + 3), p(b))
        assert np.array_equal(a, b)  # same coefficient order as the reference's C code, not only the same multiset


def test_variance(ref, oracle):
    rng = np.random.default_rng(5)
    for (w, h) in [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (4, 8), (8, 4), (16, 64), (64, 16), (128, 64)]:
        for pat in ("random", "extreme"):
            a = rng.integers(0, 256, (h, w + 1)).astype(np.uint8); b = rng.integers(0, 256, (h, w + 2)).astype(np.uint8)
            if pat == "extreme":
                a[:] = 255; b[:] = 0; b[h // 2:] = 3
            s1, s2 = C.c_uint32(), C.c_uint32()
            va = getattr(ref, f"svt_aom_variance{w}x{h}_c")(p(a), w + 1, p(b), w + 2, C.byref(s1))
            vb = oracle.orc_variance8(p(a), w + 1, p(b), w + 2, w, h, C.byref(s2))
            assert (va & 0xFFFFFFFF, s1.value) == (vb & 0xFFFFFFFF, s2.value), (w, h, pat)
            a16 = rng.integers(0, 1024, (h, w + 1)).astype(np.uint16); b16 = rng.integers(0, 1024, (h, w + 2)).astype(np.uint16)
            va = ref.svt_aom_variance_highbd_c(p(a16), w + 1, p(b16), w + 2, w, h, C.byref(s1))
            vb = oracle.orc_variance16(p(a16), w + 1, p(b16), w + 2, w, h, C.byref(s2))
            assert (va & 0xFFFFFFFF, s1.value) == (vb & 0xFFFFFFFF, s2.value), (w, h, pat)


def _qtables(rng, q):
    """zbin/round/quant/quant_shift/dequant pairs of plausible magnitude for quantizer index-like value q."""
    deq = np.array([4 + q, 4 + q + q // 3], np.int16)
    quant = np.array([(1 << 16) // max(int(d), 1) - 1 if d > 2 else 32767 for d in deq], np.int16)
    return dict(zbin=(deq * 84 // 128).astype(np.int16), round=(deq * 48 // 128).astype(np.int16), quant=quant,
                quant_shift=np.array([1 << int(rng.integers(10, 15)), 1 << int(rng.integers(10, 15))], np.int16), dequant=deq)


@pytest.mark.parametrize("highbd", [0, 1])
@pytest.mark.parametrize("use_qm", [0, 1])
def test_quantizers(ref, oracle, highbd, use_qm):
    rng = np.random.default_rng(17 + highbd + 2 * use_qm)
    for ts in (0, 1, 2, 3, 4, 5, 8, 9, 17):
        w, h = min(TX_W[ts], 32), min(TX_H[ts], 32)
        n = w * h
        log_scale = 0 if TX_W[ts] * TX_H[ts] <= 256 else (1 if TX_W[ts] * TX_H[ts] <= 1024 else 2)
        scan, iscan = np.zeros(n, np.int16), np.zeros(n, np.int16)
        ref.ref_scan_order(ts, 0, p(scan), p(iscan))
        for q in (0, 7, 60, 255, 1300):
            for pat in ("random", "zero", "dc", "minmax", "small"):
                t = _qtables(rng, q)
                lim = (1 << (15 if not highbd else 20)) - 1
                co = rng.integers(-lim, lim + 1, n).astype(np.int32)
                if pat == "zero": co[:] = 0
                if pat == "dc": co[:] = 0; co[0] = int(rng.integers(-lim, lim))
                if pat == "minmax": co[:] = np.where(rng.integers(0, 2, n) == 0, -lim, lim)
                if pat == "small": co = rng.integers(-2 * q - 8, 2 * q + 9, n).astype(np.int32)
                qm = rng.integers(16, 64, n).astype(np.uint8) if use_qm else None
                iqm = rng.integers(16, 64, n).astype(np.uint8) if use_qm else None
                pq, pi = (p(qm), p(iqm)) if use_qm else (None, None)
                res = []
                for which in ("ref", "orc"):
                    qa, da, ea = np.full(n, 7, np.int32), np.full(n, 7, np.int32), C.c_uint16(999)
                    if which == "ref":
                        fn = ref.svt_aom_highbd_quantize_b_c if highbd else ref.svt_aom_quantize_b_c_ii
                        fn(p(co), C.c_ssize_t(n), p(t["zbin"]), p(t["round"]), p(t["quant"]), p(t["quant_shift"]), p(qa), p(da), p(t["dequant"]),
                           C.byref(ea), p(scan), p(iscan), pq, pi, C.c_int32(log_scale))
                    else:
                        oracle.orc_quantize_b(p(co), C.c_ssize_t(n), p(t["zbin"]), p(t["round"]), p(t["quant"]), p(t["quant_shift"]), p(qa), p(da),
                                              p(t["dequant"]), C.byref(ea), p(scan), pq, pi, C.c_int(log_scale), C.c_int(highbd))
                    res.append((qa, da, ea.value))
                assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2], ("b", ts, q, pat)
                res = []
                for which in ("ref", "orc"):
                    qa, da, ea = np.full(n, 7, np.int32), np.full(n, 7, np.int32), C.c_uint16(999)
                    if which == "ref":
                        if use_qm:
                            fn = ref.svt_av1_highbd_quantize_fp_qm_c if highbd else ref.svt_av1_quantize_fp_qm_c
                            fn(p(co), C.c_ssize_t(n), p(t["zbin"]), p(t["round"]), p(t["quant"]), p(t["quant_shift"]), p(qa), p(da), p(t["dequant"]),
                               C.byref(ea), p(scan), p(iscan), pq, pi, C.c_int16(log_scale))
                        elif highbd:
                            ref.svt_av1_highbd_quantize_fp_c(p(co), C.c_ssize_t(n), p(t["zbin"]), p(t["round"]), p(t["quant"]), p(t["quant_shift"]), p(qa),
                                                             p(da), p(t["dequant"]), C.byref(ea), p(scan), p(iscan), C.c_int16(log_scale))
                        else:
                            fn = [ref.svt_av1_quantize_fp_c, ref.svt_av1_quantize_fp_32x32_c, ref.svt_av1_quantize_fp_64x64_c][log_scale]
                            fn(p(co), C.c_ssize_t(n), p(t["zbin"]), p(t["round"]), p(t["quant"]), p(t["quant_shift"]), p(qa), p(da), p(t["dequant"]),
                               C.byref(ea), p(scan), p(iscan))
                    else:
                        oracle.orc_quantize_fp(p(co), C.c_ssize_t(n), p(t["round"]), p(t["quant"]), p(qa), p(da), p(t["dequant"]), C.byref(ea), p(scan),
                                               pq, pi, C.c_int(log_scale), C.c_int(highbd))
                    res.append((qa, da, ea.value))
                assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2], ("fp", ts, q, pat)


def test_cos_tables_and_scan_orders(ref, oracle):
    ref.ref_cospi.restype = C.POINTER(C.c_int32)
    oracle.orc_cospi.restype = C.POINTER(C.c_int32)
    for bit in range(10, 17):
        assert np.array_equal(np.ctypeslib.as_array(ref.ref_cospi(bit), (64,)), np.ctypeslib.as_array(oracle.orc_cospi(bit), (64,)))
    for ts in range(19):
        for tt in range(16):
            a, ai, b, bi = (np.zeros(1024, np.int16) for _ in range(4))
            n = ref.ref_scan_order(ts, tt, p(a), p(ai))
            assert n == oracle.orc_scan_order(ts, tt, p(b), p(bi))
            assert np.array_equal(a[:n], b[:n]) and np.array_equal(ai[:n], bi[:n]), (ts, tt)


@pytest.mark.parametrize("ts", range(19))
def test_fwd_and_inv_txfm2d(ref, oracle, ts):
    rng = np.random.default_rng(100 + ts)
    w, h = TX_W[ts], TX_H[ts]
    wp, hp = min(w, 32), min(h, 32)
    for tt in valid_types(ts):
        for bd, pat in [(8, "random"), (10, "random"), (10, "max"), (10, "min"), (10, "checker"), (8, "laplace")]:
            stride = w + int(rng.integers(0, 5))
            r = residual_block(rng, w, h, stride, bd, pat)
            a = ref_fwd(ref, ts, tt, r, stride, bd)
            b = np.zeros(w * h, np.int32)
            oracle.orc_fwd_txfm2d(p(r), p(b), C.c_uint32(stride), tt, ts)
            assert np.array_equal(a, b), ("fwd", tt, bd, pat)
            co = a.reshape(h, w)[:hp, :wp].copy().ravel()
            if pat == "random":
                co = rng.integers(-(1 << (bd + 7)), 1 << (bd + 7), wp * hp).astype(np.int32)
            co = np.ascontiguousarray(co)
            pred = rng.integers(0, 1 << bd, (h, stride)).astype(np.uint16)
            ra = ref_inv(ref, ts, tt, co, pred, stride, bd)
            rb = pred.copy()
            oracle.orc_inv_txfm2d_add(p(co), p(pred), C.c_int32(stride), p(rb), C.c_int32(stride), tt, ts, bd)
            assert np.array_equal(ra, rb), ("inv", tt, bd, pat)


TX_NAMES = ["4x4", "8x8", "16x16", "32x32", "64x64", "4x8", "8x4", "8x16", "16x8", "16x32", "32x16", "32x64", "64x32", "4x16", "16x4", "8x32", "32x8",
            "16x64", "64x16"]


@pytest.mark.parametrize("ts", range(19))
def test_partial_frequency_transforms_are_masked_full_transforms(ref, oracle, ts):
    """av1_estimate_transform_N2 / _N4 (transforms.c:2633-2948): the reference's pruned forward transforms equal the full
    transform with everything outside the top-left quarter / sixteenth zeroed -- which is how rd_oracle.c and the HIP
    kernel implement pf_shape."""
    from txfm_cases import valid_types
    rng = np.random.default_rng(ts)
    W, H = TX_W[ts], TX_H[ts]
    nm = TX_NAMES[ts]
    for shape, sh in (("N2", 1), ("N4", 2)):
        fn = getattr(ref, f"svt_aom_transform_two_d_{nm}_{shape}_c" if W == H else f"svt_av1_fwd_txfm2d_{nm}_{shape}_c")
        for tt in valid_types(ts):
            for bd in (8, 10):
                res = rng.integers(-(1 << bd) + 1, 1 << bd, (H, W + 3)).astype(np.int16)
                full, part = np.zeros(W * H, np.int32), np.zeros(W * H, np.int32)
                oracle.orc_fwd_txfm2d(p(res), p(full), C.c_uint32(W + 3), C.c_int(tt), C.c_int(ts))
                fn(p(res), p(part), C.c_uint32(W + 3), C.c_int(tt), C.c_uint8(bd))
                f2 = full.reshape(H, W).copy()
                f2[H >> sh:, :] = 0
                f2[:, W >> sh:] = 0
                assert np.array_equal(f2.reshape(-1), part), (nm, shape, tt, bd)


@pytest.mark.parametrize("impl", ["ref", "ref_simd"])
@pytest.mark.parametrize("ts", range(5))
def test_rd_chain_oracle_equals_reference_chain(ref, oracle, ts, impl):
    """The whole tx_type_search iteration (residual -> fwd txfm -> SATD -> quantize_b -> coefficient distortion -> inv txfm
    -> SSE) through the reference's own kernels (`_c`, and the AVX2 / SSE4.1 intrinsics its x86 dispatch installs) against
    oracle/rd_oracle.c: every output of svt_hip_rd_batch, square sizes, 8 and 10 bit."""
    from svt_av1_psyex_amd import rd
    from txfm_cases import valid_types
    rng = np.random.default_rng(70 + ts)
    rows = np.stack([rd.quant_row_from_step(8, 10), rd.quant_row_from_step(60, 75), rd.quant_row_from_step(500, 640)])
    for bd in (8, 10):
        hi = (1 << bd) - 1
        dt = np.uint8 if bd == 8 else np.uint16
        for pattern in ("smooth", "random"):
            if pattern == "smooth":
                base = np.kron(rng.integers(0, hi + 1, (18, 26)).astype(np.float64), np.ones((8, 8)))[:128, :192]
                src = np.clip(base + rng.normal(0, 6 * (1 << (bd - 8)), base.shape), 0, hi).astype(dt)
                pred = np.clip(base + rng.normal(0, 3 * (1 << (bd - 8)), base.shape), 0, hi).astype(dt)
            else:
                src = rng.integers(0, hi + 1, (128, 192)).astype(dt); pred = rng.integers(0, hi + 1, (128, 192)).astype(dt)
            jobs = rd.grid_jobs(192, 128, 192, ts)
            types = [t for t in valid_types(ts) if not (impl == "ref_simd" and ts == 3 and t not in (0, 9))]  # the SSE4.1 32x32 inverse handles DCT_DCT / IDTX only
            jobs["tx_type"] = rng.choice(types, len(jobs))
            jobs["quant_row"] = rng.integers(0, 3, len(jobs))
            f = dict(bit_depth=bd, quant_kind=0, tx_size=ts, src_stride=192, pred_stride=192)
            a = pyoracle.rd_batch(f, src, pred, jobs, rows)
            b = pyoracle.rd_batch(f, src, pred, jobs, rows, impl=impl)
            for k in a:
                assert np.array_equal(a[k], b[k]), (ts, bd, pattern, k, np.argwhere(a[k] != b[k])[:3].tolist())


def test_handle_transform_semantics_against_reference(ref):
    """svt_handle_transform*_c (transforms.c:2374-2543): what tests/test_rd_gpu.py::test_leaf_handle_transform expects of the `_hip`
    entries -- energy of everything outside the kept min(W,32) x min(H,32), kept rows packed to the front for the 64-wide sizes, the rest
    of the array untouched; the _N2_N4 forms only pack."""
    rng = np.random.default_rng(55)
    for (w, h) in [(16, 64), (32, 64), (64, 16), (64, 32), (64, 64)]:
        for suf in ("", "_N2_N4"):
            fn = getattr(ref, f"svt_handle_transform{w}x{h}{suf}_c")
            fn.restype = C.c_uint64
            a = rng.integers(-(1 << 22), 1 << 22, w * h).astype(np.int32)
            m = a.reshape(h, w).astype(np.int64)
            wp, hp = min(w, 32), min(h, 32)
            want = a.copy()
            if w == 64:
                want[:wp * hp] = m[:hp, :wp].reshape(-1)
            energy = int((m ** 2).sum() - (m[:hp, :wp] ** 2).sum()) if suf == "" else 0
            got = a.copy()
            assert fn(p(got)) == energy and np.array_equal(got, want), (w, h, suf)


def test_cul_level_fwht_and_hvs_factor(oracle, ref):
    """The three small members of SURVEY 8a: svt_av1_compute_cul_level_c (full_loop.c:1449), svt_av1_fwht4x4_c (transforms.c:3099),
    get_hvs_modulation_factor (psy_rd.c:295) -- oracle == reference."""
    rng = np.random.default_rng(77)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.orc_compute_cul_level.restype = ref.svt_av1_compute_cul_level_c.restype = C.c_uint8
    for ts in (0, 2, 3, 5, 12):
        n = min(TX_W[ts], 32) * min(TX_H[ts], 32)
        scan, iscan = np.zeros(n, np.int16), np.zeros(n, np.int16)
        assert oracle.orc_scan_order(ts, 0, p(scan), p(iscan)) == n
        for trial in range(40):
            q = np.zeros(n, np.int32)
            k = int(rng.integers(0, n + 1))
            mag = int(rng.choice([1, 2, 40, 70, 5000]))
            q[scan[:k]] = rng.integers(-mag, mag + 1, k)
            if trial % 5 == 0:
                q[0] = 0
            nz = np.flatnonzero(q[scan])
            eob = np.array([nz[-1] + 1 if len(nz) else 0], np.uint16)
            assert oracle.orc_compute_cul_level(p(scan), p(q), p(eob)) == ref.svt_av1_compute_cul_level_c(p(scan), p(q), p(eob)), (ts, trial)
    for trial in range(200):
        stride = int(rng.integers(4, 12))
        lim = int(rng.choice([255, 1023, 32767]))
        src = rng.integers(-lim, lim + 1, 4 * stride).astype(np.int16)
        if trial < 3:
            src[:] = (lim, -lim, -32768)[trial]
        a, b = np.zeros(16, np.int32), np.zeros(16, np.int32)
        oracle.orc_fwht4x4(p(src), p(a), C.c_uint32(stride))
        ref.svt_av1_fwht4x4_c(p(src), p(b), C.c_uint32(stride))
        assert np.array_equal(a, b), trial
    oracle.orc_hvs_modulation_factor.restype = ref.ref_hvs_modulation_factor.restype = C.c_double
    from svt_av1_psyex_amd import api
    L = api.lib()
    L.svt_hip_hvs_modulation_factor.restype = C.c_double
    for psy in (0.0, 0.5, 1.0, 1.35, 4.0, 0.1 + 0.2):
        for isl in (0, 1):
            for tli in range(7):
                want = ref.ref_hvs_modulation_factor(C.c_double(psy), isl, tli)
                assert oracle.orc_hvs_modulation_factor(C.c_double(psy), isl, tli) == want
                assert L.svt_hip_hvs_modulation_factor(C.c_double(psy), isl, tli) == want  # host arithmetic of libsvthip.so: no GPU involved
