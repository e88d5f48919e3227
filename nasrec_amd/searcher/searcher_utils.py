"""Worker side of the search (reference: nasrec/searcher/searcher_utils.py): build the weight-sharing supernet, pin a candidate,
score it with `eval_fn`, report {choice, test_*, hash_token[, latency]}.  One worker process per GPU (searcher.py)."""
import numpy as np
import torch

from ..search_space import ops_config_lib
from ..supernet.supernet import SuperNet
from ..utils.config import NUM_EMBEDDINGS_AVAZU, NUM_EMBEDDINGS_CRITEO, NUM_EMBEDDINGS_KDD
from ..utils.io_utils import load_model_checkpoint

_num_embedding_dict = {"criteo-kaggle": NUM_EMBEDDINGS_CRITEO, "avazu": NUM_EMBEDDINGS_AVAZU, "kdd": NUM_EMBEDDINGS_KDD}
_num_sparse_inputs_dict = {"criteo-kaggle": 26, "avazu": 23, "kdd": 10}
_num_dense_inputs_dict = {"criteo-kaggle": 13, "avazu": 1, "kdd": 3}


def get_device_id(job_id, on_cpu=False):
    return None if on_cpu else job_id


def build_supernet(args, num_embeddings=None):
    """searcher_utils.py:61-69"""
    return SuperNet(sparse_input_size=_num_sparse_inputs_dict[args.dataset], num_blocks=args.num_blocks, ops_config=ops_config_lib[args.config],
                    use_layernorm=(args.use_layernorm == 1), activation="relu",
                    num_embeddings=num_embeddings if num_embeddings is not None else _num_embedding_dict[args.dataset],
                    path_sampling_strategy="full-path")


def _create_model_train_and_get_results(args, gpu_id, eval_fn, tokenizer, choice, checkpoint, kwargs):
    """searcher_utils.py:57-104"""
    args.gpu = gpu_id
    tables = getattr(args, "num_embeddings", None)
    model = build_supernet(args, tables)
    if choice is not None:
        model.configure_choice(choice)
    results = eval_fn(model, args, checkpoint)
    results["hash_token"] = tokenizer.hash_token(tokenizer.tokenize(model.choice))
    if kwargs.get("beta", 0.0) != 0.0:
        from ..utils.train_utils import get_model_latency
        cur_choice = model.choice
        del model
        n = kwargs["latency_batch_size"]
        int_x = torch.rand((n, _num_dense_inputs_dict[args.dataset]), dtype=torch.float32)
        cat_x = torch.zeros((n, _num_sparse_inputs_dict[args.dataset]), dtype=torch.int64)
        print("Getting latency of fixed model...")
        fixed = SuperNet(num_blocks=args.num_blocks, ops_config=ops_config_lib[args.config], use_layernorm=(args.use_layernorm == 1),
                         activation="relu", num_embeddings=tables if tables is not None else _num_embedding_dict[args.dataset],
                         sparse_input_size=_num_sparse_inputs_dict[args.dataset], path_sampling_strategy="fixed-path", fixed=True,
                         fixed_choice=cur_choice)
        mean_lat, _ = get_model_latency(fixed, (int_x, cat_x), gpu_id)
        results["latency"] = mean_lat
        print("Latency: {:.5f} s.".format(mean_lat))
    return results


def create_model_train_and_get_results_helper(args, gpu_id, eval_fn, tokenizer, choice, return_dict, ckpt_holder, kwargs):
    """searcher_utils.py:109-127: process entry point.  The checkpoint is read once and shared through the manager dict; every worker
    reseeds np.random from the OS so that parallel workers draw different candidates."""
    if "ckpt" not in list(ckpt_holder.keys()):
        ckpt_holder["ckpt"] = load_model_checkpoint(args.ckpt_path) if args.ckpt_path is not None else None
    checkpoint = ckpt_holder["ckpt"]
    if not getattr(args, "deterministic_workers", False):
        np.random.seed(None)
    return_dict["worker_{}".format(gpu_id)] = _create_model_train_and_get_results(args, gpu_id, eval_fn, tokenizer, choice, checkpoint, kwargs=kwargs)
