"""Checkpoint files and resume state in the reference's format.

    {label}_net_{G0,G1,...,D,D_T0,D_T1,...}.pth   plain ``state_dict`` per network
                                                  (models/utils.py:6-9; generator.py:292-294;
                                                  discriminator.py:250-253)
    iter.txt                                      "epoch,epoch_iter" written with np.savetxt(fmt='%d')
                                                  (models/models.py:62-68, :96-110)

``load_network`` keeps the reference's three-stage fallback (base_model.py:25-62): exact load, then
the keys both sides share, then the tensors whose shapes agree (the rest are reported and keep their
initial values).  Files are read with ``weights_only=True`` (nothing in a checkpoint is executed).
The state_dict key names are the reference's because the modules mirror its containers
(ir2rgb_amd.networks; pinned by tests/test_networks_cpu.py against goldens of the reference).
"""
import os

import numpy as np
import torch


def network_path(save_dir, network_label, epoch_label):
    return os.path.join(save_dir, f"{epoch_label}_net_{network_label}.pth")


def save_network(network, network_label, epoch_label, save_dir):
    """models/utils.py:6-9."""
    os.makedirs(save_dir, exist_ok=True)
    torch.save(network.state_dict(), network_path(save_dir, network_label, epoch_label))


def load_network(network, network_label, epoch_label, save_dir, log=print):
    """base_model.py:25-62.  Returns the sorted list of top-level module names that were NOT initialised
    from the file (empty on a full load).  A missing G0 file is an error, as in the reference."""
    path = network_path(save_dir, network_label, epoch_label)
    if not os.path.isfile(path):
        log(f"{path} not exists yet!")
        if "G0" in network_label:
            raise FileNotFoundError("Generator must exist!")
        return None
    pretrained = torch.load(path, map_location="cpu", weights_only=True)
    try:
        network.load_state_dict(pretrained)
        return []
    except RuntimeError:
        pass
    model = network.state_dict()
    shared = {k: v for k, v in pretrained.items() if k in model}
    try:
        network.load_state_dict(shared)
        log(f"Pretrained network {network_label} has excessive layers; Only loading layers that are used")
        return []
    except RuntimeError:
        pass
    log(f"Pretrained network {network_label} has fewer layers; The following are not initialized:")
    for k, v in pretrained.items():
        if k in model and v.size() == model[k].size():
            model[k] = v
    missing = sorted({k.split(".")[0] for k, v in model.items() if k not in pretrained or v.size() != pretrained[k].size()})
    log(missing)
    network.load_state_dict(model)
    return missing


def write_iter(save_dir, epoch, epoch_iter):
    """np.savetxt(iter_path, (epoch, epoch_iter), delimiter=',', fmt='%d') (models/models.py:103, :110)."""
    os.makedirs(save_dir, exist_ok=True)
    np.savetxt(os.path.join(save_dir, "iter.txt"), (epoch, epoch_iter), delimiter=",", fmt="%d")


def read_iter(save_dir):
    """(start_epoch, epoch_iter); (1, 0) when there is nothing to resume from (models/models.py:64-68)."""
    path = os.path.join(save_dir, "iter.txt")
    if not os.path.exists(path):
        return 1, 0
    e, i = np.loadtxt(path, delimiter=",", dtype=int)
    return int(e), int(i)


def save_trainer(trainer, label, save_dir, epoch=None, epoch_iter=None):
    """Vid2VidModelG.save + Vid2VidModelD.save (+ iter.txt when epoch is given)."""
    for s, g in enumerate(trainer.netG):
        save_network(g, f"G{s}", label, save_dir)
    save_network(trainer.netD, "D", label, save_dir)
    for s, d in enumerate(trainer.netD_T):
        save_network(d, f"D_T{s}", label, save_dir)
    if epoch is not None:
        write_iter(save_dir, epoch, epoch_iter or 0)


def load_trainer(trainer, label, save_dir, log=print):
    """continue_train / load_pretrain path of the reference (generator.py:58-62, discriminator.py:52-58).
    Parameters are updated in place, so optimizers and packed-weight caches (keyed on the version counter,
    which load_state_dict's copy_ bumps) stay valid."""
    for s, g in enumerate(trainer.netG):
        load_network(g, f"G{s}", label, save_dir, log)
    load_network(trainer.netD, "D", label, save_dir, log)
    for s, d in enumerate(trainer.netD_T):
        load_network(d, f"D_T{s}", label, save_dir, log)
    return read_iter(save_dir)
