#!/usr/bin/env python3
"""Throughput of the drop-in nn.Module route (model(int_x, cat_x) + loss.backward() + torch optimizer) on the bench workload:
normal training, and the searcher's fine-tune-last-layer mode."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nasrec_amd.supernet.supernet import SuperNet, ops_config_lib
from nasrec_amd.utils.config import NUM_EMBEDDINGS_CRITEO
import bench

ca = json.load(open(os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
m = SuperNet(num_blocks=ca["num_blocks"], ops_config=ops_config_lib[ca["config"]], use_layernorm=False, num_embeddings=NUM_EMBEDDINGS_CRITEO,
             sparse_input_size=26, path_sampling_strategy="fixed-path", fixed=True, fixed_choice=ca).cuda()
int_x, cat_x, y = bench.synthetic_batches(1, 256, 13, NUM_EMBEDDINGS_CRITEO, "cuda", 1)[0]
y = y.view(-1, 1)
with torch.no_grad():
    m(int_x, cat_x)
loss_fn = torch.nn.BCEWithLogitsLoss()
for mode in ("finetune_last_only", "forward_only"):
    if mode == "finetune_last_only":
        m.set_mode_to_finelune_last_only()
        opt = torch.optim.Adagrad([p for p in m.parameters() if p.requires_grad], lr=0.01, eps=1e-2)
    def step():
        if mode == "forward_only":
            with torch.no_grad():
                m(int_x, cat_x)
        else:
            opt.zero_grad()
            loss_fn(m(int_x, cat_x), y).backward()
            opt.step()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 300
    print("%-20s %.3f ms/step  %.0f samples/s" % (mode, dt * 1e3, 256 / dt))
