"""GPU: parity under the activation statistics of TRAINED towers (tools/parity_outliers.py has the method): outlier
channels 100-1000 x the bulk's spread with small LayerNorm gains on them, token-local massive values, row means several times the
bulk's spread, injected into the seeded towers at full depth. The folded-LayerNorm HIP path (the default), the separate-LayerNorm
HIP path and the reference's own stack as PyTorch-ROCm bf16 ops, all against the fp32 CPU oracle on the same weights and pixels."""
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

# The stated bf16 tolerance, two clauses (bench.py): rel-L2 <= 2e-2 on the bulk channels, OR not above the reference stack's
# own bf16 error on the same tensors (PyTorch-ROCm bf16 ops: library GEMM, SDPA, layer_norm) by more than 10 %, with 2.5e-2 as the hard ceiling.
# The second clause is what ViViT needs under these statistics: its final LayerNorm divides the stream's error by a small bulk spread, and the
# reference stack itself reads 2.25e-2 there (HIP 2.18e-2; profiles/r05_parity_outliers.json). Measured values: profiles/r06_parity_outliers.json.
BOUND_BULK = 2e-2
BOUND_CEILING = 2.5e-2
DEPTH = {"languagebind": 23, "dinov2": 23, "vivit": 12, "siglip": 11}


@pytest.mark.parametrize("name", ["languagebind", "dinov2", "vivit", "siglip"])
def test_full_depth_parity_with_outlier_channels_and_row_offsets(dev, name):
    import parity_outliers as P
    res = P.run(dev, encoders=(name,), scenarios=("both",), batch=16)[name]["both"]
    print(name, res)
    st = res["oracle_stream"]
    if name != "vivit":  # (ViViT's tokens are read behind its final LayerNorm, which takes the injected channel / row statistics out again)
        assert st["outlier_channel_max_abs_over_bulk_spread"] >= 500 and st["always_on_channel_median_abs_over_bulk_spread"] >= 100
        assert st["row_mean_over_bulk_spread"] >= 3.0  # the injected statistics are what the test says they are
    assert res["depth"] == DEPTH[name]
    fold, sep, ref = res["hip_ln_folded_vs_oracle"], res["hip_ln_separate_vs_oracle"], res["torch_rocm_bf16_vs_oracle"]
    # a video gives the same bits alone and as the last of a 16-video batch (whose rows the remainder launches compute)
    assert res["hip_batch_last_video_bit_equal_to_video_alone"] == {"hip_ln_folded": True, "hip_ln_separate": True}
    for key in ("rel_l2", "rel_l2_bulk"):
        assert fold[key] <= max(1.10 * ref[key], ref[key] + 5e-4), (key, fold, ref)  # not above the reference stack's own error
        assert sep[key] <= max(1.10 * ref[key], ref[key] + 5e-4), (key, sep, ref)
    # clause 1, or clause 2 under the ceiling
    assert fold["rel_l2_bulk"] <= BOUND_BULK or (fold["rel_l2_bulk"] <= 1.10 * ref["rel_l2_bulk"] and fold["rel_l2_bulk"] <= BOUND_CEILING), (fold, ref)
    assert fold["min_cos_bulk"] >= 0.999, fold
    if name != "vivit":
        assert fold["rel_l2_bulk"] <= BOUND_BULK, fold  # three of the four meet the first clause outright
