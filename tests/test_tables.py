"""Host constant tables vs the torch expressions the reference evaluates (CPU)."""
import numpy as np
import torch
import torch.nn.functional as F

from cris.pytorch_amd import tables
from oracle import cris_oracle as O


def test_pos_tables_match_reference_formulas():
    np.testing.assert_allclose(tables.pos2d_table(512, 26, 26), O.pos2d(512, 26, 26).numpy(), atol=2e-6)
    np.testing.assert_allclose(tables.pos2d_table(128, 3, 5), O.pos2d(128, 3, 5).numpy(), atol=2e-6)
    np.testing.assert_allclose(tables.pos1d_table(512, 17), O.pos1d(512, 17).numpy(), atol=2e-6)


def test_bicubic_matrix_matches_interpolate():
    for G, H, W in [(7, 13, 13), (7, 15, 15), (2, 3, 3), (7, 5, 9)]:
        R = torch.from_numpy(tables.bicubic_resize_matrix(G, H, W))
        x = torch.randn(1, 6, G, G, generator=torch.Generator().manual_seed(G + H))
        ref = F.interpolate(x, size=(H, W), mode="bicubic", align_corners=False).flatten(2)[0].t()
        got = R @ x.flatten(2)[0].t()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5)
