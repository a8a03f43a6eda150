"""Pins the plain-C oracle (oracle/tfmq_oracle_c.c: bin indices, MINMAX / MSE-candidate arithmetic, AdaRound
hard indices, integer-accumulate w4a8 conv) to the reference's golden vectors.  CPU only."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def oc():
    spec = importlib.util.spec_from_file_location("ob", os.path.join(ROOT, "oracle", "build_oracle_c.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return C.CDLL(m.build())


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def test_bin_indices_and_scalers(oc, golden):
    g = golden("f1_quantizer")
    x = np.ascontiguousarray(g["acts"], np.float32)
    for nm in ("minmax", "mse"):
        q = np.empty(x.size, np.uint8)
        oc.oc_quant_index(fp(x), C.c_size_t(x.size), C.c_float(float(g[f"acts_{nm}_delta"])), C.c_float(float(g[f"acts_{nm}_zp"])), 256, up(q))
        assert np.array_equal(q.reshape(x.shape), g[f"acts_{nm}_idx"])
    d, z = C.c_float(), C.c_float()
    oc.oc_minmax_qparam(C.c_float(float(x.min())), C.c_float(float(x.max())), 256, 0, C.byref(d), C.byref(z))
    assert d.value == float(g["acts_minmax_delta"]) and z.value == float(g["acts_minmax_zp"])
    ds, zs = np.empty(80, np.float32), np.empty(80, np.float32)
    oc.oc_mse_candidates(C.c_float(float(x.min())), C.c_float(float(x.max())), 256, 0, fp(ds), fp(zs))
    assert np.array_equal(ds, g["acts_mse_cand_delta"].astype(np.float32)) and np.array_equal(zs, g["acts_mse_cand_zp"].astype(np.float32))


def test_weight_indices_nearest_and_adaround(oc, golden):
    g = golden("f1_quantizer")
    w = np.ascontiguousarray(g["wts"], np.float32)
    d = np.ascontiguousarray(g["wts_mse_delta"].reshape(-1), np.float32)
    z = np.ascontiguousarray(g["wts_mse_zp"].reshape(-1), np.float32)
    q = np.empty(w.size, np.uint8)
    oc.oc_weight_index(fp(w), None, fp(d), fp(z), w.shape[0], w[0].size, 16, up(q))
    assert np.array_equal(q.reshape(w.shape), g["wts_mse_idx"])
    g4 = golden("f4_adaround")
    w = np.ascontiguousarray(g4["w"], np.float32)
    a = np.ascontiguousarray(g4["alphas"][-1], np.float32)
    d = np.ascontiguousarray(g4["wdelta"].reshape(-1), np.float32)
    z = np.ascontiguousarray(g4["wzp"].reshape(-1), np.float32)
    q = np.empty(w.size, np.uint8)
    oc.oc_weight_index(fp(w), fp(a), fp(d), fp(z), w.shape[0], w[0].size, 16, up(q))
    deq = d[:, None] * (q.reshape(w.shape[0], -1).astype(np.float32) - z[:, None])
    assert np.array_equal(deq.reshape(w.shape), g4["w_hard_final"])


def test_integer_conv_matches_reference_quantlayer(oc, golden):
    g = golden("f3_quantlayer")
    for tag, pad in (("conv3", 1), ("conv1", 0)):
        x, w, b = g[f"{tag}_x"], np.ascontiguousarray(g[f"{tag}_w"], np.float32), np.ascontiguousarray(g[f"{tag}_b"], np.float32)
        da, za = float(g[f"{tag}_adelta"]), float(g[f"{tag}_azp"])
        dw = np.ascontiguousarray(g[f"{tag}_wdelta"].reshape(-1), np.float32)
        zw = np.ascontiguousarray(g[f"{tag}_wzp"].reshape(-1), np.float32)
        xn = np.ascontiguousarray(x.transpose(0, 2, 3, 1), np.float32)
        qa = np.empty(xn.size, np.uint8)
        oc.oc_quant_index(fp(xn), C.c_size_t(xn.size), C.c_float(da), C.c_float(za), 256, up(qa))
        qw = np.empty(w.size, np.uint8)
        oc.oc_weight_index(fp(w), None, fp(dw), fp(zw), w.shape[0], w[0].size, 16, up(qw))
        B, H, W_, Cin = xn.shape
        Cout, _, KH, KW = w.shape
        y = np.empty((B, H, W_, Cout), np.float32)
        zwi = np.ascontiguousarray(zw.astype(np.int32))
        oc.oc_conv_w4a8(up(qa), B, H, W_, Cin, up(qw), Cout, KH, KW, 1, pad, pad, H, W_, C.c_float(da), int(za), fp(dw),
                        zwi.ctypes.data_as(C.POINTER(C.c_int)), fp(b), fp(y))
        ref = g[f"{tag}_y"].transpose(0, 2, 3, 1)
        assert np.abs(y - ref).max() / np.abs(ref).max() <= 1e-5
