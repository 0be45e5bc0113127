"""CPU: host-side constants of the SFT step against the oracle / reference formulas (no kernels): sinusoidal tables, DDPM schedule, and the
sparse matrix form of DINOv2's bicubic position-embedding resampling (forward taps and their transpose)."""
import torch

from oracle import dinov2 as o_dino
from oracle import nextdit as o_nd
from oracle.nn_ref import sinusoidal_pos_emb as o_spe
from oracle.schedulers import DDPMScheduler


def test_sinusoidal_tables_and_ddpm_schedule():
    from internnav_amd import sft as E

    t = torch.tensor([1.0, 37.0, 999.0, 1000.0])
    assert torch.equal(E.timestep_embedding(t), o_nd.timestep_embedding(t))
    assert torch.equal(E.sinusoidal_positions(32, 384, "cpu"), o_nd.sinusoidal_positional_encoding(32, 384))
    ts = torch.tensor([0, 3, 19])
    assert torch.equal(E.sinusoidal_pos_emb(ts, 384), o_spe(ts, 384))
    for n in (10, 20, 100):
        assert torch.equal(E.ddpm_alphas_cumprod(n), DDPMScheduler(num_train_timesteps=n).alphas_cumprod)


def _dense(idx, coef, n_src):
    A = torch.zeros(idx.shape[0], n_src)
    for r in range(idx.shape[0]):
        for j in range(idx.shape[1]):
            if idx[r, j] >= 0:
                A[r, idx[r, j]] += coef[r, j]
    return A


def test_pos_embed_resampling_as_sparse_rows():
    """DinoTrain builds the 37x37 -> 16x16 bicubic resampling (dinov2.py:180-211) as a sparse row mix and its transpose for the gradient."""
    from internnav_amd.sft import DinoTrain

    d = DinoTrain("rgb_model.", "cpu")
    g = torch.Generator().manual_seed(0)
    pe = torch.randn(1, 37 * 37 + 1, 384, generator=g)
    ref = o_dino.interpolate_pos_embed(pe, 224, 224)[0, 1:]                        # [256, 384]
    A = _dense(d.fwd_idx, d.fwd_coef, 37 * 37)
    assert d.fwd_idx.shape[1] <= 16 and A.shape == (256, 1369)                     # bicubic: at most 4 x 4 taps per output
    assert (A @ pe[0, 1:] - ref).abs().max().item() < 1e-5
    At = _dense(d.bwd_idx, d.bwd_coef, 256)
    assert torch.allclose(At, A.t(), atol=0, rtol=0)                                # the gradient path applies exactly A^T
    assert abs(A.sum(1) - 1).max().item() < 1e-5                                    # partition of unity of the cubic convolution kernel
