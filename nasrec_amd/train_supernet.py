"""Weight-sharing supernet training — the engine-side counterpart of the reference's nasrec/train_supernet.py (same flags,
defaults, logging-directory naming, log pickle and checkpoint), for scripts/train_supernet/*.sh:

    python -u nasrec_amd/train_supernet.py --root_dir ./data/criteo_kaggle_autoctr --config xlarge --num_blocks 7 \\
        --use_layernorm 1 --strategy default --anypath_choice binomial-0.5 --supernet_training_steps 15000 ...

`--root_dir synthetic[:steps=N,...]` substitutes dataset-shaped random batches; no TensorBoard graph dump (see main_train.py)."""
import argparse
import os
import sys

sys.path.append(os.getcwd())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from nasrec_amd.main_train import _num_embedding_dict, _num_sparse_inputs_dict, build_lr_scheduler, build_optimizer, summary_writer  # noqa: E402
from nasrec_amd.supernet.supernet import SuperNet, ops_config_lib  # noqa: E402
from nasrec_amd.utils.data_pipes import make_loaders  # noqa: E402
from nasrec_amd.utils.io_utils import create_dir, dump_pickle_data, load_model_checkpoint, load_optimizer_state, save_model_checkpoint  # noqa: E402
from nasrec_amd.utils.train_utils import (get_l2_loss, get_model_flops_and_params, init_weights, train_and_test_one_epoch,  # noqa: E402
                                          warmup_supernet_model)


def train_and_eval_one_model(model, args):
    train_loader, test_loader = make_loaders(args)
    with torch.no_grad():
        model = warmup_supernet_model(model, train_loader, args.gpu, freeze_gc=True)  # this process trains this one model
    flops, params = get_model_flops_and_params(model, train_loader, args.gpu)
    print("FLOPS: {:.4f} M \t Params: {:.4f} M".format(flops / 1e6, params / 1e6))
    model.configure_path_sampling_strategy(args.strategy)
    if args.loss_function != "bce":
        raise NotImplementedError("Loss function {} is not implemented!".format(args.loss_function))
    loss_fn = torch.nn.BCEWithLogitsLoss()

    def l2_loss_fn(m):
        return get_l2_loss(m, args.wd, args.no_reg_param_name, gpu=args.gpu)

    optimizer = build_optimizer(args.optimizer, model, args.learning_rate)
    steps_per_epoch = args.train_limit // args.train_batch_size
    num_train_steps = steps_per_epoch * args.num_epochs
    lr_scheduler = build_lr_scheduler(args.lr_schedule, optimizer, num_train_steps, num_train_steps // 10, args.learning_rate)
    if args.checkpoint_path is not None:
        checkpoint = load_model_checkpoint(args.checkpoint_path)
        model.load_state_dict(checkpoint["model_state_dict"], strict=True)
        if "optimizer_state_dict" in checkpoint:
            load_optimizer_state(model, optimizer, checkpoint["optimizer_state_dict"])
    else:
        model.apply(init_weights)
    from nasrec_amd.utils.dist import assert_replicas_identical, broadcast_replica_state
    broadcast_replica_state(model)  # data parallel: rank 0's weights, tables and accumulators are THE model
    assert_replicas_identical(model)
    print(model)
    strategy_name = args.strategy if args.strategy == "single-path" else args.strategy + "-" + args.anypath_choice
    logging_dir = os.path.join(args.logging_dir, "supernet_{}blocks_layernorm{:d}_{}_lr{:.2f}_supernetwarmup_{}".format(
        args.num_blocks, args.use_layernorm, strategy_name, args.learning_rate, args.supernet_training_steps))
    print("Logging in directory: {}".format(logging_dir))
    create_dir(logging_dir)
    writer = summary_writer(args.logging_dir)
    epoch_logs, logs = [], None
    for epoch in range(args.num_epochs):
        logs = train_and_test_one_epoch(
            model, epoch, optimizer, lr_scheduler, train_loader, test_loader, loss_fn, l2_loss_fn, args.train_batch_size, args.gpu,
            display_interval=args.display_interval, test_interval=args.test_interval, max_train_steps=steps_per_epoch,
            test_only_at_last_step=True, grad_clip_value=5.0, tb_writer=writer)
        epoch_logs.append(logs)
    print("Dumping logs to {}!".format(logging_dir))
    from nasrec_amd.utils.dist import world_info
    if world_info()[0] == 0:  # replicas are identical (broadcast at start, same global-batch update on every rank): rank 0 writes the artefacts
        dump_pickle_data(os.path.join(logging_dir, "train_test_logs.pickle"), logs)  # the last epoch only, as the reference does
        save_model_checkpoint(model, os.path.join(logging_dir, "supernet_checkpoint.pt"), optimizer)
    elif getattr(model, "_table_sharding", None):  # row-sharded tables: the whole-table state_dict is a collective
        save_model_checkpoint(model, None, optimizer)  # (tables and their accumulators are gathered: every rank takes part)
    return epoch_logs


def main(args):
    from nasrec_amd.utils.dist import init_from_env
    rank, world = init_from_env(args)  # torchrun: one process per GPU, args.gpu = the local rank; same path seed on every rank
    model = SuperNet(sparse_input_size=_num_sparse_inputs_dict[args.dataset], num_blocks=args.num_blocks,
                     ops_config=ops_config_lib[args.config], use_layernorm=(args.use_layernorm == 1), activation="relu",
                     num_embeddings=_num_embedding_dict[args.dataset], path_sampling_strategy=args.strategy,
                     anypath_choice=args.anypath_choice, supernet_training_steps=args.supernet_training_steps, candidate_choices=None,
                     table_sharding=args.table_sharding)
    return train_and_eval_one_model(model.to(args.gpu), args)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", type=str, default="criteo-kaggle", help="Choice of datasets", choices=["criteo-kaggle", "avazu", "kdd"])
    p.add_argument("--root_dir", type=str, default=None, help="Root Directory for dataset.")
    p.add_argument("--logging_dir", type=str, default=None, help="Directory to put loggings.")
    p.add_argument("--strategy", type=str, default="single-path", help="Path sampling strategy",
                   choices=["evo-2shot-path", "default", "single-path", "any-path", "full-path", "fixed-path"])
    p.add_argument("--use_layernorm", type=int, default=0, help="Whether use layernorm or not.")
    p.add_argument("--config", type=str, default="xlarge", help="Search space configuration.")
    p.add_argument("--num_blocks", type=int, default=7, help="Number of blocks per supernet.")
    p.add_argument("--checkpoint_path", type=str, default=None, help="Checkpoint to resume from.")
    p.add_argument("--evo_2shot_path_candidates", type=str, default=None, help="2nd shot path sampling candidates.")
    p.add_argument("--wd", type=float, default=0, help="L2 Weight decay")
    p.add_argument("--learning_rate", type=float, default=0.01, help="Learning rate")
    p.add_argument("--learning_rate_decay", type=float, default=0, help="Learning rate decay.")
    p.add_argument("--num_epochs", type=int, default=1, help="Number of epochs for training.")
    p.add_argument("--supernet_training_steps", type=int, default=2000, help="Steps of full-supernet warm-up with a decaying probability.")
    p.add_argument("--anypath_choice", type=str, default="uniform", help="Distribution of the number of active nodes under any-path.")
    p.add_argument("--train_batch_size", type=int, default=200, help="Training batch size.")
    p.add_argument("--test_batch_size", type=int, default=16368, help="Testing batch size.")
    p.add_argument("--train_limit", type=int, default=36672495, help="Maximum number of training examples.")
    p.add_argument("--test_limit", type=int, default=6548659, help="Maximum number of testing examples.")
    p.add_argument("--lr_schedule", default="cosine", help="Learning rate schedule", choices=["cosine", "constant", "constant-no-warmup"])
    p.add_argument("--display_interval", type=int, default=100, help="Interval to display training stats.")
    p.add_argument("--test_interval", type=int, default=100, help="Testing intervals.")
    p.add_argument("--activation", type=str, default="relu", help="Activation function.", choices=["relu", "silu"])
    p.add_argument("--train_split", type=str, default="train", choices=["train", "trainval"])
    p.add_argument("--validate_split", type=str, default="test", choices=["val", "test"])
    p.add_argument("--no-reg-param-name", type=str, default=None, help="Name of the parameters that do not need to be regularized.")
    p.add_argument("--loss_function", type=str, default="bce", choices=["bce"])
    p.add_argument("--optimizer", type=str, default="adagrad", choices=["adagrad", "sgd", "adam", "rmsprop", "ds-optimizer"])
    p.add_argument("--pretrained_dlrm_emb_path", type=str, default=None, help="Pretrained embedding path from DLRM model.")
    p.add_argument("--gpu", type=int, default=0, help="GPU ID to use.")
    # not a reference flag: placement of the embedding tables under torchrun (nasrec_amd/sharded_tables.py)
    p.add_argument("--table-sharding", dest="table_sharding", type=str, default="none", choices=["none", "row"],
                   help="none: whole tables on every rank; row: every rank owns a row range of every table (all-to-all of ids / rows / row gradients)")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
