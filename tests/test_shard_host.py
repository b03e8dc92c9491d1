"""Packed shard file format and loader bookkeeping (host side; the device collate is tested in test_gpu_shard.py)."""
import numpy as np
import pytest
import torch

from cartnet_amd import shard
from cartnet_amd.data import DataLoader
from cartnet_amd.synthetic import make_crystal


def _items(adp=True):
    return [make_crystal(300 + g, n, adp=adp) for g, n in enumerate((5, 9, 1, 14, 3))]


def test_file_round_trip_is_exact_and_aligned(tmp_path):
    items = _items()
    arrays = shard.pack(items)
    path = str(tmp_path / "train.cnshard")
    shard.write_shard(path, items)
    back = shard.read_shard(path)
    assert set(back) == set(arrays)
    for k, a in arrays.items():
        assert back[k].dtype == a.dtype and back[k].shape == a.shape and np.array_equal(back[k], a), k
        assert back[k].offset % 64 == 0
    assert arrays["atom_ptr"].tolist() == [0, 5, 14, 15, 29, 32]
    assert arrays["y"].shape[1] == 9 and arrays["y_ptr"][-1] == sum(int(d.non_H_mask.sum()) for d in items)
    # edge indices are stored relative to their crystal
    for g, d in enumerate(items):
        lo, hi = arrays["edge_ptr"][g], arrays["edge_ptr"][g + 1]
        assert np.array_equal(arrays["edge_src"][lo:hi], d.edge_index[0].numpy())
        assert np.array_equal(arrays["edge_tgt"][lo:hi], d.edge_index[1].numpy())


def test_scalar_target_shard_has_no_mask_or_temperature():
    arrays = shard.pack(_items(adp=False))
    assert "non_h_mask" not in arrays and "temperature" not in arrays
    assert arrays["y"].shape == (5, 1) and arrays["y_ptr"].tolist() == [0, 1, 2, 3, 4, 5]


def test_rejects_bad_input(tmp_path):
    p = tmp_path / "x.bin"
    p.write_bytes(b"not a shard at all")
    with pytest.raises(ValueError):
        shard.read_shard(str(p))
    with pytest.raises(ValueError):
        shard.pack([])
    d = _items()[1]
    d.edge_index = d.edge_index.flip(1)
    with pytest.raises(ValueError):
        shard.pack([d])
    with pytest.raises(ValueError):
        shard.DeviceShard(shard.pack(_items()), device="cpu")          # no CPU path


def test_loader_order_matches_the_host_dataloader():
    class _FakeShard:
        num_graphs, device = 23, torch.device("cpu")
        edge_ptr = np.concatenate([[0], np.cumsum([10 + (7 * i) % 13 for i in range(23)])]).astype(np.int64)

    class _Item:
        def __init__(self, i):
            self.edge_index = torch.zeros((2, 10 + (7 * i) % 13), dtype=torch.int64)
    for world in (1, 2):
        for rank in range(world):
            a = shard.ShardLoader.__new__(shard.ShardLoader)
            a.shard, a.batch_size, a.shuffle, a.seed, a.rank, a.world_size = _FakeShard, 4, True, 5, rank, world
            a.drop_last, a.indices, a.epoch = False, list(range(23)), 3
            b = DataLoader([_Item(i) for i in range(23)], 4, shuffle=True, seed=5, rank=rank, world_size=world)
            b.epoch = 3
            assert a._batches() == b._batches() and len(a) == len(b)
