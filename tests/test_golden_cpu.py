"""Oracle vs the committed golden fixtures (CPU).  The fixtures were produced by the fp64
restatement; here BOTH restatements are re-run against them so a drift in either is caught."""
import numpy as np
import pytest

from oracle import gnf_oracle as O
from helpers import ATTN_GOLDEN_CASES, BN_GOLDEN_CASES, DATA_DRIVER_GOLDEN_CASES, GOLDEN_CASES, load_golden

# attention and batch-norm fixtures run through the same checks
GOLDEN_CASES = GOLDEN_CASES + ATTN_GOLDEN_CASES + BN_GOLDEN_CASES + DATA_DRIVER_GOLDEN_CASES


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_fp64_oracle_reproduces_golden(name):
    g = load_golden(name)
    n = int(g["n_node"].sum())
    o = O.Fp64Dense(g["senders"], g["receivers"], n, agg=g["agg"], combine=g["combine"],
                    epsilon=g["epsilon"], activation=g["activation"])
    res = o.log_prob(g["x"], g["params"], g["T"], g["weight_sharing"])
    np.testing.assert_allclose(res["z"], g["z"], rtol=0, atol=1e-12)
    assert abs(res["log_det_jacobian"] - float(g["logdet"])) < 1e-10
    assert abs(res["log_prob_xs_per_node"] - float(g["log_prob_xs_per_node"])) < 1e-12
    np.testing.assert_allclose(o.g(g["z"], g["params"], g["T"], g["weight_sharing"]), g["x_roundtrip"], atol=1e-10)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_fp32_oracle_matches_golden(name):
    g = load_golden(name)
    n = int(g["n_node"].sum())
    o = O.Fp32Gather(g["senders"], g["receivers"], n, agg=g["agg"], combine=g["combine"],
                     epsilon=g["epsilon"], activation=g["activation"])
    res = o.log_prob(o.to_t(g["x"]), o.prep_params(g["params"]), g["T"], g["weight_sharing"])
    assert abs(res["log_prob_xs_per_node"] - float(g["log_prob_xs_per_node"])) < 1e-5
    np.testing.assert_allclose(res["z"].numpy(), g["z"], atol=5e-5, rtol=1e-5)
