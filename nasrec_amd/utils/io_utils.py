"""Small file helpers of the training harness (reference: nasrec/utils/io_utils.py): JSON / pickle round trips, directory
creation and the checkpoint container `{"model_state_dict": ..., "optimizer_state_dict": ...}` (io_utils.py:59-79) that
`train_supernet.py --checkpoint_path` and `eval_subnet_from_supernet.py` read back."""
import json
import os
import pickle

import torch


def _need(name, what):
    assert name is not None, "%s should not be 'None'!" % what
    return name


def load_json(json_file_name=None):
    with open(_need(json_file_name, "Json file name"), "r") as fp:
        return json.load(fp)


def dump_json(json_file_name, data):
    with open(_need(json_file_name, "Json file name"), "w") as fp:
        json.dump(data, fp)


def create_dir(dir_name=None):
    os.makedirs(_need(dir_name, "Directory name"), exist_ok=True)


def dump_pickle_data(dump_path, data):
    with open(_need(dump_path, "Dump path"), "wb") as fp:
        pickle.dump(data, fp)


def load_pickle_data(load_path):
    with open(_need(load_path, "Load path"), "rb") as fp:
        return pickle.load(fp)


def load_model_checkpoint(load_path):
    """-> dict on the CPU (the caller moves tensors with load_state_dict)."""
    print("Loading weights from {}!".format(_need(load_path, "Load model path")))
    return torch.load(load_path, map_location=torch.device("cpu"))


def save_model_checkpoint(model, save_path, optimizer=None):
    """save_path None: build the checkpoint but write nothing — row-sharded tables make `model.state_dict()` a collective (every rank
    contributes its rows), so every rank calls this and only rank 0 passes a path"""
    blob = {"model_state_dict": model.state_dict()}
    if optimizer is not None:
        blob["optimizer_state_dict"] = optimizer.state_dict()
    if save_path is None:
        return
    torch.save(blob, save_path)
    print("Saved weights to {}!".format(save_path))
