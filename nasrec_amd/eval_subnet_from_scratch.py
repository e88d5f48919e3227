"""Train sampled (or cached) sub-networks from scratch and record their scores — the engine-side counterpart of the reference's
nasrec/eval_subnet_from_scratch.py (the "ground truth" leg of the ranking study: scripts/train_subnet/*.sh run it on 100 random
sub-networks per search space), with the same flags, defaults, printouts and `results.pickle`:

    python -u nasrec_amd/eval_subnet_from_scratch.py --learning_rate 0.12 --gpu 0 --root_dir ./data/criteo_kaggle_autoctr \\
        --train_batch_size 1024 --logging_dir ./www-test/criteo-100subnets-xlarge-b0 --test_batch_size 8000 --use_layernorm 0 \\
        --config xlarge --num_blocks 7 --num_subnets 100

Each sub-network is a FIXED `SuperNet` (its path drawn once by `fixed-path` from the global `np.random` stream seeded with
`--random_seed`, or taken from `--choice_from_pickle_file`), initialised with `init_weights`, trained for `--num_epochs` through
`train_and_test_one_epoch` — the fused engine step whenever the recipe allows it (Adagrad, weight decay 0), the torch route otherwise —
and tested once at the end of every epoch (eval_subnet_from_scratch.py:150-170).  A diverged model (test AUROC < 0) is skipped, as the
reference does.  `--root_dir synthetic[:steps=N,...]` trains on dataset-shaped random batches."""
import argparse
import os
import sys
import warnings

sys.path.append(os.getcwd())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from nasrec_amd.main_train import _num_embedding_dict, _num_sparse_inputs_dict, build_lr_scheduler, build_optimizer  # noqa: E402
from nasrec_amd.supernet.supernet import SuperNet, ops_config_lib  # noqa: E402
from nasrec_amd.utils.data_pipes import make_loaders  # noqa: E402
from nasrec_amd.utils.io_utils import create_dir, dump_pickle_data, load_pickle_data  # noqa: E402
from nasrec_amd.utils.train_utils import (get_l2_loss, get_model_flops_and_params, init_weights, train_and_test_one_epoch,  # noqa: E402
                                          warmup_model)

warnings.simplefilter("ignore", ResourceWarning)


def train_and_eval_one_model(model, args):
    """eval_subnet_from_scratch.py:67-171 -> one log dict per epoch (with the model's MACs and parameter count in it)"""
    train_loader, test_loader = make_loaders(args)
    with torch.no_grad():
        model = warmup_model(model, train_loader, args.gpu)
    flops, params = get_model_flops_and_params(model, train_loader, args.gpu)
    print("FLOPS: {:.4f} M \t Params: {:.4f} M".format(flops / 1e6, params / 1e6))
    if args.loss_function != "bce":
        raise NotImplementedError("Loss function {} is not implemented!".format(args.loss_function))
    loss_fn = torch.nn.BCEWithLogitsLoss()
    model.apply(init_weights)
    optimizer = build_optimizer(args.optimizer, model, args.learning_rate)
    steps_per_epoch = args.train_limit // args.train_batch_size
    # (the warm-up shrinks with the number of epochs here — :134 — unlike main_train.py)
    lr_scheduler = build_lr_scheduler(args.lr_schedule, optimizer, steps_per_epoch * args.num_epochs, steps_per_epoch // 10 // args.num_epochs,
                                      args.learning_rate)
    epoch_logs = []
    for epoch in range(args.num_epochs):
        logs = train_and_test_one_epoch(
            model, epoch, optimizer, lr_scheduler, train_loader, test_loader, loss_fn,
            lambda m: get_l2_loss(m, args.wd, args.no_reg_param_name, gpu=args.gpu), args.train_batch_size, args.gpu,
            test_interval=args.test_interval, max_train_steps=steps_per_epoch if args.max_train_steps == -1 else args.max_train_steps,
            max_eval_steps=args.max_eval_steps, test_only_at_last_step=True, grad_clip_value=5.0)
        logs["flops(M)"], logs["Params(M)"] = flops / 1e6, params / 1e6
        epoch_logs.append(logs)
    return epoch_logs


def main(args):
    np.random.seed(args.random_seed)  # the candidates come from the global stream (supernet.py `fixed-path`)
    if args.choice_from_pickle_file is not None:
        all_choices = load_pickle_data(args.choice_from_pickle_file)
        num_subnets = len(all_choices)
        print("Evaluating {} subnets from record file: {}".format(num_subnets, args.choice_from_pickle_file))
    else:
        all_choices, num_subnets = None, args.num_subnets
    all_results = []
    create_dir(args.logging_dir)
    tables = getattr(args, "num_embeddings", None) or _num_embedding_dict[args.dataset]
    for i in range(num_subnets):
        choice = None if all_choices is None else all_choices[i]["choice"]
        print("Evaluating {:d} out of {:d} subnetworks!".format(i, num_subnets))
        model = SuperNet(sparse_input_size=_num_sparse_inputs_dict[args.dataset], num_blocks=args.num_blocks, ops_config=ops_config_lib[args.config],
                         use_layernorm=(args.use_layernorm == 1), activation=args.activation, num_embeddings=tables,
                         path_sampling_strategy="fixed-path", fixed=True, fixed_choice=choice)
        model = model.to(args.gpu)
        logs = train_and_eval_one_model(model, args)
        print("Trained model with the following choice...")
        print(model.choice)
        if logs[0]["test_AUROC"][-1] < 0:
            print("Model Diverged! Skip logging...")
            continue
        all_results.append({"choice": model.choice, "test_acc": logs[0]["test_Accuracy"][-1], "test_auroc": logs[0]["test_AUROC"][-1],
                            "test_loss": logs[0]["test_loss"][-1]})
        dump_pickle_data(os.path.join(args.logging_dir, "results.pickle"), all_results)
        del model
        torch.cuda.empty_cache()
    return all_results


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", type=str, default="criteo-kaggle", choices=["criteo-kaggle", "avazu", "kdd"], help="Choice of datasets")
    p.add_argument("--root_dir", type=str, default=None, help="Root Directory for dataset.")
    p.add_argument("--logging_dir", type=str, default=None, help="Directory to put loggings.")
    p.add_argument("--max_train_steps", type=int, default=-1, help="Maximum steps to train. '-1' to train a whole epoch.")
    p.add_argument("--max_eval_steps", type=int, default=-1, help="Maximum steps to evaluate. '-1' to evaluate a whole epoch.")
    p.add_argument("--use_layernorm", type=int, default=0, help="Whether use layernorm in the supernet or not.")
    p.add_argument("--config", type=str, default="xlarge", help="Configuration for the supernet search.")
    p.add_argument("--num_blocks", type=int, default=7, help="Number of blocks per supernet.")
    p.add_argument("--random_seed", type=int, default=None, help="Random seed to carry sampling.")
    p.add_argument("--wd", type=float, default=0, help="L2 Weight decay")
    p.add_argument("--learning_rate", type=float, default=0.01, help="Learning rate")
    p.add_argument("--learning_rate_decay", type=float, default=0, help="Learning rate decay.")
    p.add_argument("--num_epochs", type=int, default=1, help="Number of epochs for training.")
    p.add_argument("--train_batch_size", type=int, default=200, help="Training batch size.")
    p.add_argument("--test_batch_size", type=int, default=16368, help="Testing batch size.")
    p.add_argument("--lr_schedule", default="cosine", choices=["cosine", "constant", "constant-no-warmup"], help="Learning rate schedule")
    p.add_argument("--train_limit", type=int, default=36672495, help="Maximum number of training examples.")
    p.add_argument("--test_limit", type=int, default=4296061, help="Maximum number of testing examples.")
    p.add_argument("--test_interval", type=int, default=100, help="Testing interval when training supernet.")
    p.add_argument("--train_split", type=str, default="train", choices=["train", "trainval"])
    p.add_argument("--validate_split", type=str, default="val", choices=["val", "test"])
    p.add_argument("--activation", type=str, default="relu", choices=["relu", "silu"])
    p.add_argument("--num_subnets", type=int, default=20, help="Number of experimental subnets.")
    p.add_argument("--choice_from_pickle_file", type=str, default=None,
                   help="Pre-sampled cached choices from a pickle file; the number of records overrides 'num_subnets'.")
    p.add_argument("--no-reg-param-name", type=str, default=None, help="Name of the parameters that do not need to be regularized.")
    p.add_argument("--optimizer", type=str, default="adagrad", choices=["adagrad", "sgd", "adam", "rmsprop"])
    p.add_argument("--loss_function", type=str, default="bce", choices=["bce"])
    p.add_argument("--gpu", type=int, default=0, help="GPU ID to use.")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
