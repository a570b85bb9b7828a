"""CPU check of the MFMA register-layout algebra the deformation kernels rely on (csrc/deform.hip): a lane-level numpy
model of v_mfma_f32_32x32x2_f32 / v_mfma_f32_4x4x1_16b_f32 replays DenseTrunk, DenseIL (interleaved activations), DenseT
(transposed products), the dW2 block through the transposed LDS tile and the k<=4 small-head forms, and compares every
product with plain matrix algebra.  (The instructions themselves are exercised by the -m gpu parity tests.)"""
import importlib.util
import os

import numpy as np
import pytest

_spec = importlib.util.spec_from_file_location(
    "mfma_layout_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mfma_layout_model.py"))
model = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(model)


@pytest.mark.parametrize("WT,FCH,k", [(4, 4, 3), (4, 4, 48), (4, 6, 4), (2, 8, 1), (2, 16, 48), (4, 12, 3), (2, 4, 4)])
def test_layouts_consistent(WT, FCH, k):
    assert model.check(WT, FCH, k, np.random.default_rng(WT * 100 + FCH * 10 + k))


def test_mfma_models_match_definition():
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(64), rng.standard_normal(64)
    d = model.mfma32(a, b, np.zeros((64, 16)))
    # D[i][j] = sum_k A[i][k] B[k][j], A[i][k] in lane i+32k, B[k][j] in lane j+32k, D in lane j+32*hh register r, i = rho(r,hh)
    for r in range(16):
        for hh in range(2):
            i = model.rho(r, hh)
            for j in (0, 7, 31):
                assert np.isclose(d[j + 32 * hh, r], a[i] * b[j] + a[i + 32] * b[j + 32])
    d4 = model.mfma4(a, b, np.zeros((64, 4)))
    assert np.isclose(d4[4 * 5 + 2, 3], a[4 * 5 + 3] * b[4 * 5 + 2])


@pytest.mark.parametrize("W,F,k", [(128, 32, 3), (128, 32, 48), (64, 64, 4), (128, 48, 1), (64, 32, 48), (128, 96, 4)])
def test_16_gaussian_forms_consistent(W, F, k):
    """The v_mfma_f32_16x16x4_f32 forms of the forward kernel (csrc/deform_fwd16.h): one wave = 16 Gaussians."""
    assert model.check16(W, F, k, np.random.default_rng(W + F + k))


def test_mfma16_model_matches_definition():
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal(64), rng.standard_normal(64)
    d = model.mfma16(a, b, np.zeros((64, 4)))
    for (n, q, r) in ((0, 0, 0), (5, 2, 3), (15, 3, 1)):
        row = 4 * q + r
        assert np.isclose(d[n + 16 * q, r], sum(a[row + 16 * kk] * b[n + 16 * kk] for kk in range(4)))
