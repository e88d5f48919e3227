"""Train and evaluate one model — the engine-side counterpart of the reference's nasrec/main_train.py with the same flags,
defaults, printouts, log pickle and checkpoint, so the published recipes (scripts/eval_best_model/*.sh) run verbatim:

    python -u nasrec_amd/main_train.py --root_dir ./data/criteo_kaggle_autoctr/ --net supernet-config \\
        --supernet_config nasrec_amd/configs/criteo/ea_criteo_kaggle_xlarge_best_1shot.json --num_epochs 1 \\
        --learning_rate 0.16 --train_batch_size 256 --wd 0 --logging_dir ./experiments/... --gpu 0 --test_interval 10000

Differences: `--root_dir synthetic[:steps=N,...]` generates dataset-shaped random batches (the datasets cannot be shipped);
the TensorBoard graph dump (a jit trace of the model, main_train.py:129-137) is skipped — the network is one launch plan,
not a traceable module graph — and the scalar writer is used only if tensorboard is installed."""
import argparse
import json
import os
import sys
import warnings

sys.path.append(os.getcwd())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from nasrec_amd.supernet.modules import flags  # noqa: E402
from nasrec_amd.supernet.supernet import SuperNet, ops_config_lib  # noqa: E402
from nasrec_amd.utils.config import NUM_EMBEDDINGS_AVAZU, NUM_EMBEDDINGS_CRITEO, NUM_EMBEDDINGS_KDD  # noqa: E402
from nasrec_amd.utils.data_pipes import make_loaders  # noqa: E402
from nasrec_amd.utils.io_utils import create_dir, dump_pickle_data, load_json, save_model_checkpoint  # noqa: E402
from nasrec_amd.utils.lr_schedule import ConstantWithWarmup, CosineAnnealingWarmupRestarts  # noqa: E402
from nasrec_amd.utils.train_utils import (get_l2_loss, get_model_flops_and_params, init_weights, train_and_test_one_epoch,  # noqa: E402
                                          warmup_model)

warnings.simplefilter("ignore", ResourceWarning)
warnings.simplefilter("ignore", UserWarning)

_num_sparse_inputs_dict = {"criteo-kaggle": 26, "avazu": 23, "kdd": 10}
_num_embedding_dict = {"criteo-kaggle": NUM_EMBEDDINGS_CRITEO, "avazu": NUM_EMBEDDINGS_AVAZU, "kdd": NUM_EMBEDDINGS_KDD}


def summary_writer(logging_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(logging_dir)
    except Exception:  # tensorboard not installed
        return None


def build_optimizer(name, model, lr):
    """main_train.py:150-160 (Adagrad eps 1e-2; Adam eps 1e-8; SGD nesterov momentum 0.9)"""
    if name == "adagrad":
        return torch.optim.Adagrad(model.parameters(), lr=lr, eps=1e-2)
    if name == "adam":
        return torch.optim.Adam(model.parameters(), lr=lr, eps=1e-8)
    if name == "sgd":
        return torch.optim.SGD(model.parameters(), lr=lr, nesterov=True, momentum=0.9)
    raise KeyError(name)


def build_lr_scheduler(kind, optimizer, num_train_steps, num_warmup_steps, max_lr):
    if kind == "cosine":
        return CosineAnnealingWarmupRestarts(optimizer, first_cycle_steps=num_train_steps, warmup_steps=num_warmup_steps, max_lr=max_lr,
                                             min_lr=1e-8)
    if kind == "constant":
        return ConstantWithWarmup(optimizer, num_warmup_steps=num_warmup_steps)
    return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[num_train_steps * 10], gamma=0.1)  # "constant-no-warmup"


def train_and_eval_one_model(model, args):
    train_loader, test_loader = make_loaders(args)
    with torch.no_grad():
        model = warmup_model(model, train_loader, args.gpu)
    flops, params = get_model_flops_and_params(model, train_loader, args.gpu)
    print("FLOPS: {:.4f} M \t Params: {:.4f} M".format(flops / 1e6, params / 1e6))
    if args.loss_function != "bce":
        raise NotImplementedError("Loss function {} is not implemented!".format(args.loss_function))
    loss_fn = torch.nn.BCEWithLogitsLoss()
    writer = summary_writer(args.logging_dir)
    flags.config_debug(False)

    def l2_loss_fn(m):
        return get_l2_loss(m, args.wd, args.no_reg_param_name, gpu=args.gpu)

    optimizer = build_optimizer(args.optimizer, model, args.learning_rate)
    steps_per_epoch = args.train_limit // args.train_batch_size
    num_train_steps = steps_per_epoch * args.num_epochs
    num_warmup_steps = steps_per_epoch // 10 // args.num_epochs
    lr_scheduler = build_lr_scheduler(args.lr_schedule, optimizer, num_train_steps, num_warmup_steps, args.learning_rate)
    model.apply(init_weights)
    from nasrec_amd.utils.dist import assert_replicas_identical, broadcast_replica_state
    broadcast_replica_state(model)  # data parallel: rank 0's weights, tables and accumulators are THE model
    assert_replicas_identical(model)
    print(model)
    create_dir(args.logging_dir)
    with open(os.path.join(args.logging_dir, "configs_args.json"), "w") as f:
        json.dump(args.__dict__, f, indent=2)
    epoch_logs = []
    for epoch in range(args.num_epochs):
        # epoch index 1 is special-cased by the reference: 1000 steps, a test every 100 (main_train.py:206-207)
        logs = train_and_test_one_epoch(
            model, epoch, optimizer, lr_scheduler, train_loader, test_loader, loss_fn, l2_loss_fn, args.train_batch_size, args.gpu,
            display_interval=args.display_interval, test_interval=args.test_interval if epoch != 1 else 100,
            max_train_steps=steps_per_epoch if epoch != 1 else 1000, test_only_at_last_step=(args.test_only_at_last_step) == 1,
            tb_writer=writer, use_amp=False, grad_clip_value=5.0)
        epoch_logs.append(logs)
    print("Dumping logs to {}!".format(args.logging_dir))
    from nasrec_amd.utils.dist import world_info
    ckpt = os.path.join(args.logging_dir, "{}_checkpoint.pt".format(args.net))
    if world_info()[0] == 0:  # replicas are identical (broadcast at start, same global-batch update on every rank): rank 0 writes the artefacts
        save_model_checkpoint(model, ckpt, optimizer)
        dump_pickle_data(os.path.join(args.logging_dir, "train_test_logs.pickle"), epoch_logs)
    elif getattr(model, "_table_sharding", None):  # row-sharded tables: the whole-table state_dict is a collective
        save_model_checkpoint(model, None, optimizer)  # (tables and their accumulators are gathered: every rank takes part)
    return epoch_logs


def get_model(args):
    """main_train.py:233-272"""
    if args.net == "supernet":
        return SuperNet(sparse_input_size=_num_sparse_inputs_dict[args.dataset], num_blocks=7, ops_config=ops_config_lib["xlarge"],
                        use_layernorm=True, activation=args.activation, num_embeddings=_num_embedding_dict[args.dataset],
                        path_sampling_strategy="full-path", table_sharding=args.table_sharding)
    if args.net == "supernet-config":
        choice = load_json(args.supernet_config)
        print(choice)
        # the reference builds best-1shot sub-networks WITHOUT LayerNorm whatever the JSON says (main_train.py:262)
        return SuperNet(sparse_input_size=_num_sparse_inputs_dict[args.dataset], num_blocks=choice["num_blocks"],
                        ops_config=ops_config_lib[choice["config"]], use_layernorm=False, activation=args.activation,
                        num_embeddings=_num_embedding_dict[args.dataset], path_sampling_strategy="fixed-path", fixed=True,
                        fixed_choice=choice, table_sharding=args.table_sharding)
    raise NotImplementedError("Model {} is not implemented!".format(args.net))


def main(args):
    from nasrec_amd.utils.dist import init_from_env
    rank, world = init_from_env(args)  # torchrun: one process per GPU, args.gpu = the local rank
    create_dir(args.logging_dir)
    model = get_model(args).to(args.gpu)
    return train_and_eval_one_model(model, args)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", type=str, default="criteo-kaggle", help="Choice of datasets", choices=["criteo-kaggle", "avazu", "kdd"])
    p.add_argument("--root_dir", type=str, default="/home/tunhouzhang/local/datasets/criteo_kaggle_sharded")
    p.add_argument("--logging_dir", type=str, default=None, help="Directory to put loggings.")
    p.add_argument("--net", type=str, default="dlrm", help="Network backbone name", choices=["supernet", "supernet-config"])
    p.add_argument("--supernet_config", type=str, default=None, help="Supernet configuration.")
    p.add_argument("--wd", type=float, default=1e-8, help="L2 Weight decay")
    p.add_argument("--learning_rate", type=float, default=0.01, help="Learning rate")
    p.add_argument("--learning_rate_decay", type=float, default=0, help="Learning rate decay.")
    p.add_argument("--num_epochs", type=int, default=1, help="Number of epochs for training.")
    p.add_argument("--train_split", type=str, default="trainval", choices=["train", "trainval"],
                   help="Data split for training. Can be one of ['train', 'trainval']")
    p.add_argument("--validate_split", type=str, default="test", choices=["val", "test"],
                   help="Data split for validation (evaluation). Can be one of ['val', 'test']")
    p.add_argument("--train_batch_size", type=int, default=200, help="Training batch size.")
    p.add_argument("--test_batch_size", type=int, default=16368, help="Testing batch size.")
    p.add_argument("--optimizer", type=str, default="adagrad", help="Optimizer", choices=["adagrad", "sgd", "adam", "rmsprop", "ds-optimizer"])
    # Criteo: train 36672495 / val 4584061 / test 4584061 / trainval 41256556; Avazu: 32343175 / 4042896 / 4042896 / 36386071;
    # KDD: 119711284 / 14963910 / 14963910 / 134675194
    p.add_argument("--train_limit", type=int, default=41256556, help="Maximum number of training examples.")
    p.add_argument("--test_limit", type=int, default=4584061, help="Maximum number of testing examples.")
    p.add_argument("--lr_schedule", default="cosine", help="Learning rate schedule", choices=["cosine", "constant", "constant-no-warmup"])
    p.add_argument("--display_interval", type=int, default=100, help="Interval to display tensorboard curve/training stats.")
    p.add_argument("--test_interval", type=int, default=2000, help="Testing intervals.")
    p.add_argument("--activation", type=str, default="relu", help="Activation function to use in this work.", choices=["relu", "silu"])
    p.add_argument("--no-reg-param-name", type=str, default=None, help="Name of the parameters that do not need to be regularized.")
    p.add_argument("--loss_function", type=str, default="bce", help="Loss function to perform the task.", choices=["bce"])
    p.add_argument("--test_only_at_last_step", type=int, default=0, help="Whether only test the last step.")
    p.add_argument("--ema", type=float, default=0.0, help="EMA strength ranging from 0 to 1 (not implemented, as in the reference).")
    p.add_argument("--gpu", type=int, default=None, help="GPU ID to use.")
    # not a reference flag: placement of the embedding tables under torchrun (nasrec_amd/sharded_tables.py)
    p.add_argument("--table-sharding", dest="table_sharding", type=str, default="none", choices=["none", "row"],
                   help="none: every rank holds whole tables (replicated, row gradients all-gathered); row: every rank owns a row range "
                        "of every table, ids / rows / row gradients travel by all-to-all (tables that outgrow one GPU).  Needs the fused "
                        "step: --optimizer adagrad --wd 0")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
