"""The compressed artefact: raw native-endian float32 files weight-{l}-{out}-{in} / bias-{l}-{n}
(reference utils/ModelSave.py:8-62; byte-compatible with the authors' BRIEF_CUDA decoder)."""
import os
import shutil

import numpy as np
import torch


def save_model(model, save_path, devive="cpu"):
    """utils/ModelSave.py:32-52"""
    if not hasattr(model, "net"):
        torch.save(model.state_dict(), save_path)
        return
    if os.path.exists(save_path):
        shutil.rmtree(save_path)
    os.mkdir(save_path)
    for l in range(len(model.net)):
        weight = model.net[l][0].weight.data.to("cpu")
        bias = model.net[l][0].bias.data.to("cpu")
        np.ascontiguousarray(weight.numpy(), dtype=np.float32).tofile(
            os.path.join(save_path, "weight-{}-{}-{}".format(l, weight.shape[0], weight.shape[1])))
        np.ascontiguousarray(bias.numpy(), dtype=np.float32).tofile(os.path.join(save_path, "bias-{}-{}".format(l, len(bias))))


def load_model(model, model_path, device="cpu"):
    """utils/ModelSave.py:8-30"""
    if not hasattr(model, "net"):
        model.load_state_dict(torch.load(model_path))
        return model
    for file in os.listdir(model_path):
        file_path = os.path.join(model_path, file)
        if "weight" in file:
            _, l, shape0, shape1 = file.split("-")
            l, shape0, shape1 = int(l), int(shape0), int(shape1)
            weight = np.fromfile(file_path, dtype=np.float32, count=shape0 * shape1).reshape(shape0, shape1)
            model.net[l][0].weight.data = torch.tensor(weight).to(device)
        elif "bias" in file:
            _, l, length = file.split("-")
            l, length = int(l), int(length)
            model.net[l][0].bias.data = torch.tensor(np.fromfile(file_path, dtype=np.float32, count=length)).to(device)
    return model


def CopyDir(old_dir, new_dir):
    """utils/ModelSave.py:54-62"""
    if not os.path.exists(new_dir):
        os.mkdir(new_dir)
    for file in os.listdir(old_dir):
        shutil.copy(os.path.join(old_dir, file), os.path.join(new_dir, file))
