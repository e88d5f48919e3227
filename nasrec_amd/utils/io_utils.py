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


def _embedding_state_slots(model, optimizer):
    """[(index of the parameter in optimizer.state_dict()["state"], field f)] for the row-sharded tables of `model`"""
    pos = {id(p): i for i, p in enumerate(p for g in optimizer.param_groups for p in g["params"])}
    out = []
    for name, p in model.named_parameters():
        if name.startswith("_embedding.") and id(p) in pos:
            out.append((pos[id(p)], int(name.split(".")[1])))
    return out


def optimizer_state_for_checkpoint(model, optimizer):
    """optimizer.state_dict(), except that with row-sharded tables (SuperNet(table_sharding="row"), world > 1) the Adagrad accumulator of
    every table is returned WHOLE ([num_embeddings, 16], all-gather of the ranks' shards: a collective, every rank calls it) — like
    `model.state_dict()` does for the tables themselves — so that a checkpoint holds every rank's accumulators in the reference's
    shapes, whatever the placement that wrote it."""
    sd = optimizer.state_dict()
    sh = getattr(model, "_sharded", None)
    if getattr(model, "_table_sharding", None) == "row" and sh is not None and sh.world > 1:
        state = dict(sd["state"])
        for i, f in _embedding_state_slots(model, optimizer):
            if i in state and "sum" in state[i]:
                st = dict(state[i])
                st["sum"] = sh.whole_table(f, "state")
                state[i] = st
        sd = {"state": state, "param_groups": sd["param_groups"]}
    return sd


def load_optimizer_state(model, optimizer, state_dict):
    """optimizer.load_state_dict for a checkpoint written by save_model_checkpoint: with row-sharded tables every rank keeps rows
    [lo, hi) of the whole-table accumulators (a checkpoint of ANY world size fits; one whose accumulators are neither whole nor this
    rank's shard is refused instead of being loaded onto other rows)."""
    if getattr(model, "_table_sharding", None) == "row":
        state = dict(state_dict["state"])
        for i, f in _embedding_state_slots(model, optimizer):
            if i not in state or "sum" not in state[i]:
                continue
            lo, hi, rows = model._shard_rows(model._num_embeddings[f])
            t = state[i]["sum"]
            if int(t.shape[0]) == int(model._num_embeddings[f]) and (hi - lo != int(t.shape[0]) or lo != 0):
                part = t[lo:hi]
                if hi - lo < rows:
                    part = torch.cat([part, torch.zeros(rows - (hi - lo), t.shape[1], dtype=t.dtype, device=t.device)])
                st = dict(state[i])
                st["sum"] = part.clone()
                state[i] = st
            elif int(t.shape[0]) != rows:
                raise ValueError("optimizer checkpoint: the accumulator of table %d has %d rows; expected the whole table (%d) or this rank's shard (%d)"
                                 % (f, int(t.shape[0]), int(model._num_embeddings[f]), rows))
        state_dict = {"state": state, "param_groups": state_dict["param_groups"]}
    optimizer.load_state_dict(state_dict)


def save_model_checkpoint(model, save_path, optimizer=None):
    """save_path None: build the checkpoint but write nothing — row-sharded tables make `model.state_dict()` and the optimizer's
    table accumulators collectives (every rank contributes its rows), so every rank calls this with the same arguments and only
    rank 0 passes a path"""
    blob = {"model_state_dict": model.state_dict()}
    if optimizer is not None:
        blob["optimizer_state_dict"] = optimizer_state_for_checkpoint(model, optimizer)
    if save_path is None:
        return
    torch.save(blob, save_path)
    print("Saved weights to {}!".format(save_path))
