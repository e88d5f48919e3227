"""Score sub-networks of a trained supernet on the engine (counterpart of nasrec/eval_subnet_from_supernet.py): random search,
regularized evolution, or a cached list of choices.  A candidate is scored by `finetune_and_eval_one_model`: warm the weight-
sharing supernet up on the full path, pin the candidate's path, load the supernet checkpoint, fine-tune — by default ONLY the
last layer (`set_mode_to_finelune_last_only`, eval_subnet_from_supernet.py:114-120) — for `max_train_steps` steps and evaluate
`max_eval_steps` batches.  On the engine the last-layer mode launches just the weight part of the final-logit backward
(`bwd_final_only`), so a candidate costs a forward plus one small launch per step; candidates are independent, one per GPU
process (`--num_parallel_workers`)."""
import argparse
import os

import torch

from .search_space import ops_config_lib
from .searcher.searcher import Searcher
from .searcher.searcher_utils import _num_embedding_dict, _num_sparse_inputs_dict, build_supernet
from .utils.data_pipes import make_loaders
from .utils.io_utils import create_dir, dump_pickle_data, load_model_checkpoint, load_pickle_data
from .utils.lr_schedule import ConstantWithWarmup, CosineAnnealingWarmupRestarts
from .utils.train_utils import get_l2_loss, train_and_test_one_epoch, warmup_supernet_model


def finetune_and_eval_one_model(model, args, checkpoint):
    """eval_subnet_from_supernet.py:71-207 -> {"choice", "test_acc", "test_auroc", "test_loss"}"""
    train_loader, test_loader = make_loaders(args)
    model = model.to(args.gpu)
    with torch.no_grad():
        model = warmup_supernet_model(model, train_loader, args.gpu)  # full path: every lazy shape, every deleted projection
    model.configure_path_sampling_strategy("fixed-path")
    if args.loss_function != "bce":
        raise NotImplementedError("Loss function {} is not implemented!".format(args.loss_function))
    loss_fn = torch.nn.BCEWithLogitsLoss()
    if args.finetune_whole_supernet == 0:
        print("Finetune last only ...")
        model.set_mode_to_finelune_last_only()
    else:
        print("Finetuning the whole supernet.")

    def l2_loss_fn(m):
        return get_l2_loss(m, args.wd, getattr(args, "no_reg_param_name", None), gpu=args.gpu)

    if args.optimizer == "adagrad":
        optimizer = torch.optim.Adagrad(model.parameters(), lr=args.learning_rate, eps=1e-2)
    elif args.optimizer == "adam":
        optimizer = torch.optim.Adam(model.parameters(), lr=args.learning_rate, eps=1e-8)
    elif args.optimizer == "sgd":
        optimizer = torch.optim.SGD(model.parameters(), lr=args.learning_rate, nesterov=True, momentum=0.9)
    else:
        raise KeyError(args.optimizer)
    num_train_steps = args.max_train_steps * args.num_epochs
    num_warmup_steps = args.max_train_steps // 10
    if args.lr_schedule == "cosine":
        lr_scheduler = CosineAnnealingWarmupRestarts(optimizer, first_cycle_steps=num_train_steps, warmup_steps=num_warmup_steps,
                                                     max_lr=args.learning_rate, min_lr=1e-8)
    elif args.lr_schedule == "constant":
        lr_scheduler = ConstantWithWarmup(optimizer, num_warmup_steps=num_warmup_steps)
    elif args.lr_schedule == "stepwise":
        lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[num_train_steps // 3, num_train_steps * 2 // 3], gamma=0.2)
    else:
        lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[num_train_steps * 10], gamma=0.1)
    if checkpoint is not None:
        model.load_state_dict(checkpoint["model_state_dict"], strict=True)  # (the optimizer state is deliberately not restored, :173-177)
    lr_scheduler.step(epoch=-1)  # back to the first step after reading the state_dict (:179)
    logs = train_and_test_one_epoch(model, 0, optimizer, lr_scheduler, train_loader, test_loader, loss_fn, l2_loss_fn, args.train_batch_size,
                                    args.gpu, max_train_steps=args.max_train_steps, max_eval_steps=args.max_eval_steps,
                                    test_interval=max(2, args.max_train_steps), test_only_at_last_step=(args.test_only_at_last_step == 1),
                                    grad_clip_value=5.0)
    torch.cuda.empty_cache()
    return {"choice": model.choice, "test_acc": logs["test_Accuracy"], "test_auroc": logs["test_AUROC"], "test_loss": logs["test_loss"]}


def main(args):
    print("Logging dir: {}".format(args.logging_dir))
    create_dir(args.logging_dir)
    if args.method == "random":
        searcher = Searcher(finetune_and_eval_one_model, args)
        all_results = searcher.random_search_from_supernet(
            args.random_budget, num_parallel_workers=args.num_parallel_workers, beta=args.beta, target_latency=args.target_latency,
            latency_batch_size=args.latency_batch_size, sorted=(args.random_sort_results == 1), top_k=args.random_search_topk,
            criterion=args.criterion)
    elif args.method == "regularized-ea":
        searcher = Searcher(finetune_and_eval_one_model, args)
        all_results = searcher.regularized_evolution_from_supernet(
            num_parallel_workers=args.num_parallel_workers, sample_size=args.sample_size, init_population=args.init_population,
            n_childs=args.n_childs, n_generations=args.n_generations, beta=args.beta, target_latency=args.target_latency,
            latency_batch_size=args.latency_batch_size, top_k=args.ea_top_k, criterion=args.criterion)
    elif args.method == "cached":
        assert args.choice_from_pickle_file is not None, \
            "'--choice_from_pickle_file' should not be None if you want to train from cached records!"
        all_choices = load_pickle_data(args.choice_from_pickle_file)
        print("Evaluating {} subnets from record file: {}".format(len(all_choices), args.choice_from_pickle_file))
        all_results = []
        checkpoint = load_model_checkpoint(args.ckpt_path)
        for idx, rec in enumerate(all_choices):
            print("Evaluating {} of {} networks!".format(idx, len(all_choices)))
            print("GT performance: {:.5f}".format(rec["test_loss"]))
            print(rec["choice"])
            model = build_supernet(args, getattr(args, "num_embeddings", None))
            model.configure_choice(rec["choice"])
            all_results.append(finetune_and_eval_one_model(model.to(args.gpu), args, checkpoint))
    else:
        raise NotImplementedError("Method {} not supported!".format(args.method))
    dump_pickle_data(os.path.join(args.logging_dir, "results.pickle"), all_results)
    return all_results


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--num_blocks", type=int, default=7, help="Number of blocks in a supernet.")
    p.add_argument("--config", type=str, default="xlarge", help="Configuration for the supernet search.")
    p.add_argument("--ckpt_path", type=str, default=None, help="Path to the checkpoint.")
    p.add_argument("--max_train_steps", type=int, default=-1, help="Maximum steps to train. '-1' to train a whole epoch.")
    p.add_argument("--max_eval_steps", type=int, default=-1, help="Maximum steps to evaluate. '-1' to evaluate a whole epoch.")
    p.add_argument("--method", type=str, default="cached", help="Search method.", choices=["random", "regularized-ea", "cached"])
    p.add_argument("--dataset", type=str, default="criteo-kaggle", choices=["criteo-kaggle", "avazu", "kdd"])
    p.add_argument("--root_dir", type=str, default="r")
    p.add_argument("--logging_dir", type=str, default=None)
    p.add_argument("--train_split", type=str, default="train", choices=["train", "trainval"])
    p.add_argument("--validate_split", type=str, default="val", choices=["val", "test"])
    p.add_argument("--choice_from_pickle_file", type=str, default=None)
    p.add_argument("--use_layernorm", type=int, default=0)
    p.add_argument("--finetune_whole_supernet", type=int, default=0)
    p.add_argument("--wd", type=float, default=0)
    p.add_argument("--learning_rate", type=float, default=0.01)
    p.add_argument("--sparse_lr", type=float, default=0.003)
    p.add_argument("--learning_rate_decay", type=float, default=0)
    p.add_argument("--num_epochs", type=int, default=1)
    p.add_argument("--lr_schedule", default="cosine", choices=["cosine", "constant", "constant-no-warmup", "stepwise"])
    p.add_argument("--train_batch_size", type=int, default=200)
    p.add_argument("--test_batch_size", type=int, default=16368)
    p.add_argument("--train_limit", type=int, default=36672495)
    p.add_argument("--test_limit", type=int, default=6548659)
    p.add_argument("--activation", type=str, default="relu", choices=["relu", "silu"])
    p.add_argument("--no-reg-param-name", type=str, default=None)
    p.add_argument("--loss_function", type=str, default="bce", choices=["bce"])
    p.add_argument("--test_only_at_last_step", type=int, default=0)
    p.add_argument("--display_interval", type=int, default=100)
    p.add_argument("--gpu", type=int, default=None)
    p.add_argument("--num_parallel_workers", type=int, default=1)
    p.add_argument("--random_search_topk", type=int, default=5)
    p.add_argument("--random_budget", type=int, default=5)
    p.add_argument("--random_sort_results", type=int, default=0)
    p.add_argument("--n_childs", type=int, default=8)
    p.add_argument("--n_generations", type=int, default=50)
    p.add_argument("--init_population", type=int, default=8)
    p.add_argument("--ea_top_k", type=int, default=5)
    p.add_argument("--sample_size", type=int, default=6)
    p.add_argument("--optimizer", type=str, default="adagrad", choices=["adagrad", "sgd", "adam"])
    p.add_argument("--criterion", type=str, default="test_loss")
    p.add_argument("--beta", type=float, default=0.0)
    p.add_argument("--target_latency", type=float, default=-1)
    p.add_argument("--latency_batch_size", type=int, default=512)
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
