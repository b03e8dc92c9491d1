"""BASELINE.json configs[0] plumbing: the main.py-equivalent trains CartNet (2 layers, dim 64) on 32 synthetic crystals
of ~50 atoms through the reference's call order (loaders -> create_model -> Adam/OneCycle -> train -> best checkpoint ->
reload -> test), on the GPU path."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_main_trains_and_checkpoints(tmp_path, monkeypatch):
    import main as entry
    monkeypatch.chdir(tmp_path)
    res = entry.main(["--synthetic", "32", "--atoms", "30", "70", "--dim_in", "64", "--num_layers", "2", "--epochs", "4",
                      "--batch", "4", "--batch_accumulation", "2", "--name", "plumbing", "--augment", "--lr", "2e-3"])
    hist = res["history"]
    assert len(hist) == 4 and all(torch.isfinite(torch.tensor(h["train_mae"])) for h in hist)
    assert hist[-1]["train_mae"] < hist[0]["train_mae"]            # it learns
    ck = torch.load(tmp_path / "results" / "plumbing" / "0" / "ckpt" / "best.ckpt")
    assert set(ck) == {"model_state", "optimizer_state"}            # train/train.py:92-95
    assert "encoder.embedding.weight" in ck["model_state"] and set(ck["optimizer_state"]) == {"state", "param_groups"}      # torch.optim.Adam's layout
    assert float(ck["optimizer_state"]["state"][0]["step"]) > 0
    assert "test_mae" in res and res["params"] == sum(v.numel() for k, v in ck["model_state"].items()
                                                      if "running" not in k and "num_batches" not in k and "rbf" not in k)
    tm = res["test_metrics"]                                         # train/metrics.py:201-214 on the GPU
    assert set(tm) == {"mae", "volume_percentage_error", "similarity_index", "iou"}
    assert 0.0 < tm["iou"] <= 1.0 and tm["similarity_index"] >= 0.0 and tm["volume_percentage_error"] >= 0.0


def test_main_with_resident_dataset_matches_host_loader(tmp_path, monkeypatch):
    """--resident_dataset builds the same batches on the GPU (cartnet_collate), so training is reproduced exactly."""
    import main as entry
    monkeypatch.chdir(tmp_path)
    common = ["--synthetic", "20", "--atoms", "10", "30", "--dim_in", "32", "--num_layers", "2", "--epochs", "2",
              "--batch", "4", "--batch_accumulation", "2"]
    a = entry.main(common + ["--name", "host"])
    b = entry.main(common + ["--name", "resident", "--resident_dataset"])
    assert [h["train_mae"] for h in a["history"]] == [h["train_mae"] for h in b["history"]]
    assert a["test_metrics"] == b["test_metrics"]
    c = entry.main(common + ["--name", "resident_aug", "--resident_dataset", "--augment"])
    assert all(torch.isfinite(torch.tensor(h["train_mae"])) for h in c["history"])


def test_main_runs_icomformer(tmp_path, monkeypatch):
    import main as entry
    monkeypatch.chdir(tmp_path)
    res = entry.main(["--synthetic", "12", "--atoms", "10", "20", "--dim_in", "32", "--epochs", "2", "--batch", "3",
                      "--batch_accumulation", "1", "--name", "icf", "--model", "icomformer"])
    assert len(res["history"]) == 2 and res["history"][-1]["train_mae"] == res["history"][-1]["train_mae"]


def test_main_at_width_256_with_bf16x3_gemms(tmp_path, monkeypatch):
    """dim_in = 256 takes the DMA-fed kernels; --gemm_precision 1 must train to the same loss curve as precision 0
    within the parity budget amplified by a few optimiser steps."""
    import main as entry
    monkeypatch.chdir(tmp_path)
    common = ["--synthetic", "16", "--atoms", "20", "40", "--dim_in", "256", "--num_layers", "2", "--epochs", "2",
              "--batch", "4", "--batch_accumulation", "1", "--lr", "1e-3"]
    a = entry.main(common + ["--name", "p0"])
    b = entry.main(common + ["--name", "p1", "--gemm_precision", "1"])
    for ha, hb in zip(a["history"], b["history"]):
        assert abs(ha["train_mae"] - hb["train_mae"]) < 1e-3 * abs(ha["train_mae"])


def test_main_jarvis_style_run_with_bf16_storage(tmp_path, monkeypatch):
    """BASELINE configs[2] through the CLI: scalar head (a non-ADP dataset name), no temperature, --gemm_precision 2
    --bf16_storage.  Trains, stays finite, and follows the fp32-storage bf16 run to bf16 accuracy."""
    import main as entry
    monkeypatch.chdir(tmp_path)
    common = ["--synthetic", "24", "--atoms", "2", "20", "--dim_in", "256", "--num_layers", "2", "--epochs", "2",
              "--batch", "8", "--batch_accumulation", "1", "--lr", "1e-3", "--dataset", "jarvis", "--gemm_precision", "2"]
    a = entry.main(common + ["--name", "p2"])
    b = entry.main(common + ["--name", "p2h", "--bf16_storage"])
    for ha, hb in zip(a["history"], b["history"]):
        assert hb["train_mae"] == hb["train_mae"]                                  # finite
        assert abs(ha["train_mae"] - hb["train_mae"]) < 5e-2 * abs(ha["train_mae"])


def test_main_runs_ecomformer(tmp_path, monkeypatch):
    import main as entry
    monkeypatch.chdir(tmp_path)
    res = entry.main(["--synthetic", "12", "--atoms", "10", "20", "--dim_in", "32", "--epochs", "2", "--batch", "3",
                      "--batch_accumulation", "1", "--name", "ecf", "--model", "ecomformer"])
    assert len(res["history"]) == 2 and res["history"][-1]["train_mae"] == res["history"][-1]["train_mae"]


def test_main_fused_accumulation_reproduces_the_micro_batch_recipe(tmp_path, monkeypatch):
    """--fused_accumulation: batch x accumulation micro-batches per optimiser step as one pass with per-micro-batch
    BatchNorm and loss == the reference recipe run micro-batch by micro-batch (same loaders, same order)."""
    import main as entry
    monkeypatch.chdir(tmp_path)
    common = ["--synthetic", "40", "--atoms", "10", "30", "--dim_in", "64", "--num_layers", "2", "--epochs", "2",
              "--batch", "4", "--batch_accumulation", "4", "--lr", "1e-3"]
    a = entry.main(common + ["--name", "micro"])
    b = entry.main(common + ["--name", "fused", "--fused_accumulation"])
    for ha, hb in zip(a["history"], b["history"]):
        assert abs(ha["train_mae"] - hb["train_mae"]) < 2e-3 * abs(ha["train_mae"])
        assert abs(ha["val_mae"] - hb["val_mae"]) < 2e-2 * abs(ha["val_mae"])


def test_inference_and_montecarlo_on_a_checkpoint(tmp_path, monkeypatch):
    """main.py:21-119, 212-225: --inference writes the pickle with pred / true / per-atom IoU, MAE, similarity index;
    --montecarlo rotates cart_dir and reports how far pred(rotated) is from R^T pred R (one pickle per round)."""
    import pickle
    import main as entry
    monkeypatch.chdir(tmp_path)
    common = ["--synthetic", "20", "--atoms", "10", "30", "--dim_in", "64", "--num_layers", "2"]
    entry.main(common + ["--epochs", "1", "--batch", "4", "--batch_accumulation", "1", "--name", "ck"])
    ck = str(tmp_path / "results" / "ck" / "0" / "ckpt" / "best.ckpt")
    res = entry.main(common + ["--inference", "--checkpoint_path", ck, "--inference_output", str(tmp_path / "inf.pkl")])
    assert 0.0 < res["iou_mean"] <= 1.0 and res["mae_mean"] > 0
    out = pickle.load(open(tmp_path / "inf.pkl", "rb"))
    assert len(out["pred"]) == len(out["true"]) == len(out["iou"]) == 2            # 10 % of 20 crystals, batch size 1
    assert out["pred"][0].shape == out["true"][0].shape and out["pred"][0].shape[1:] == (3, 3)
    assert out["atoms"][0].dtype == torch.int64 and out["atoms"][0].shape[0] == out["pred"][0].shape[0]
    mc = entry.main(common + ["--montecarlo", "--montecarlo_rounds", "3", "--checkpoint_path", ck,
                              "--inference_output", str(tmp_path / "inf.pkl")])
    assert mc["rounds"] == 3 and 0.0 <= mc["iou_mean"] <= 1.0 and mc["mae_mean"] >= 0
    assert all(os.path.exists(tmp_path / f"inf_montecarlo_{i}.pkl") for i in range(3))
    with pytest.raises(AssertionError, match="Weights not provided"):
        entry.main(common + ["--inference"])
