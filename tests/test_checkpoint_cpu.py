"""CPU tests of the reference-format checkpoint helpers and the VideoSeq window slicer."""
import os

import numpy as np
import pytest
import torch

from ir2rgb_amd import checkpoint as CK
from ir2rgb_amd import networks as N

OPT = dict(gen_blocks=2, n_blocks_local=1, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)


def _g(ngf=64, blocks=2, seed=0):
    torch.manual_seed(seed)
    return N.build_generator_module(9, 3, 6, ngf, "composite", 2, "batch", 0, **dict(OPT, gen_blocks=blocks))


def test_save_and_load_round_trip_uses_reference_file_names(tmp_path):
    a, b = _g(seed=1), _g(seed=2)
    CK.save_network(a, "G0", "latest", str(tmp_path))
    assert os.path.isfile(tmp_path / "latest_net_G0.pth")              # models/utils.py:7
    assert CK.load_network(b, "G0", "latest", str(tmp_path), log=lambda *_: None) == []
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    # the file is a plain state_dict with the reference's key names
    sd = torch.load(tmp_path / "latest_net_G0.pth", weights_only=True)
    assert "model_down_seg.1.weight" in sd and "model_final_flow.1.weight" in sd and "model_final_w.1.bias" in sd


def test_missing_files_follow_the_reference(tmp_path):
    d = N.build_discriminator_module(6, 64, 3, "batch", 2, True)
    assert CK.load_network(d, "D", "latest", str(tmp_path), log=lambda *_: None) is None   # "not exists yet!"
    with pytest.raises(FileNotFoundError):
        CK.load_network(_g(), "G0", "latest", str(tmp_path), log=lambda *_: None)          # "Generator must exist!"


def test_partial_loads_fall_back_like_the_reference(tmp_path):
    # (1) file has MORE layers than the model: only the shared keys are used
    big, small = _g(blocks=4, seed=3), _g(blocks=2, seed=4)
    CK.save_network(big, "G0", "10", str(tmp_path))
    logs = []
    assert CK.load_network(small, "G0", "10", str(tmp_path), log=logs.append) == []
    assert any("excessive layers" in str(m) for m in logs)
    assert torch.equal(small.state_dict()["model_down_seg.1.weight"], big.state_dict()["model_down_seg.1.weight"])
    # (2) file has FEWER layers: shape-compatible tensors are taken, the rest reported by top-level name
    small2, big2 = _g(blocks=2, seed=5), _g(blocks=4, seed=6)
    keep = {k: v.clone() for k, v in big2.state_dict().items()}
    CK.save_network(small2, "G0", "11", str(tmp_path))
    logs = []
    missing = CK.load_network(big2, "G0", "11", str(tmp_path), log=logs.append)
    assert missing and all(isinstance(m, str) for m in missing)
    sd = big2.state_dict()
    for k, v in small2.state_dict().items():
        if k in sd and sd[k].shape == v.shape:
            assert torch.equal(sd[k], v), k
    untouched = [k for k in sd if k not in small2.state_dict()]
    assert untouched and all(torch.equal(sd[k], keep[k]) for k in untouched)


def test_iter_file_round_trip(tmp_path):
    assert CK.read_iter(str(tmp_path)) == (1, 0)
    CK.write_iter(str(tmp_path), 7, 123)
    assert open(tmp_path / "iter.txt").read().split() == ["7", "123"]   # np.savetxt(fmt='%d'): one value per line
    assert CK.read_iter(str(tmp_path)) == (7, 123)
    e, i = np.loadtxt(tmp_path / "iter.txt", delimiter=",", dtype=int)  # what the reference's init_params does
    assert (int(e), int(i)) == (7, 123)
