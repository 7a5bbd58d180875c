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


def test_mask_to_rows_checks_ids_as_fancy_indexing_does():
    """Ids reach the kernels as raw row numbers: negative ones wrap as in the reference's
    ``processed_X[mask]`` (model/SGNN.py:412-415), anything outside [-n, n) raises IndexError
    there and must raise here, and an empty selection is refused."""
    from sgnn_amd.stacked import _mask_to_rows
    cpu = torch.device("cpu")
    assert _mask_to_rows(torch.tensor([-1, 0, -5]), cpu, n=5).tolist() == [4, 0, 0]
    assert _mask_to_rows(torch.tensor([4, 0]), cpu, n=5).tolist() == [4, 0]
    with pytest.raises(IndexError):
        _mask_to_rows(torch.tensor([0, 5]), cpu, n=5)
    with pytest.raises(IndexError):
        _mask_to_rows(torch.tensor([-6, 1]), cpu, n=5)
    with pytest.raises(IndexError):
        _mask_to_rows(torch.tensor([True, False, True]), cpu, n=5)          # wrong length
    with pytest.raises(ValueError):
        _mask_to_rows(torch.zeros(0, dtype=torch.int64), cpu, n=5)
    with pytest.raises(ValueError):
        _mask_to_rows(torch.zeros(5, dtype=torch.bool), cpu, n=5)
    ref = torch.arange(5.0)
    idx = torch.tensor([-1, 2, -5])
    assert torch.equal(ref[_mask_to_rows(idx, cpu, n=5)], ref[idx])
