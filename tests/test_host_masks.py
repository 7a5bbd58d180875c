"""Training masks above the boundary: the reference fancy-indexes with whatever the
loader hands over (model/SGNN.py:412-415) — a boolean mask for the Planetoid graphs, an
index array for the OGB ones (data_loader.py:133)."""
import pytest
import torch


def test_mask_to_rows_accepts_boolean_masks_and_index_arrays():
    from sgnn_amd.stacked import _mask_to_rows
    cpu = torch.device("cpu")
    m = torch.tensor([False, True, True, False, True])
    assert _mask_to_rows(m, cpu).tolist() == [1, 2, 4]
    assert _mask_to_rows(m.to(torch.uint8), cpu).tolist() == [1, 2, 4]        # byte tensors index as masks
    idx = torch.tensor([4, 0, 3])
    assert _mask_to_rows(idx, cpu).tolist() == [4, 0, 3]                      # ids as given, order kept
    assert _mask_to_rows(idx.to(torch.int32), cpu).dtype == torch.int64
    assert _mask_to_rows([True, False, True], cpu).tolist() == [0, 2]
    with pytest.raises(TypeError):
        _mask_to_rows(torch.tensor([0.0, 1.0]), cpu)
    with pytest.raises(ValueError):
        _mask_to_rows(torch.zeros(2, 2, dtype=torch.int64), cpu)
