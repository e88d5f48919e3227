import sys, ctypes as C; sys.path.insert(0,'/root/repo')
import torch
from nasrec_amd import _lib as L
from nasrec_amd.utils.config import NUM_EMBEDDINGS_CRITEO as T
from bench import time_desc
lib = L.load(); st = torch.cuda.Stream(); sp = st.cuda_stream
B, Fs = 256, 26
g = torch.Generator().manual_seed(0)
def run(tables, label):
    idx = torch.stack([torch.randint(0, int(t), (B,), generator=g) for t in tables], 1).cuda()
    dout = torch.randn(B, Fs, 16).cuda(); leader = torch.zeros(B*Fs, dtype=torch.int32).cuda(); gsum = torch.zeros(B*Fs*16).cuda(); part = torch.zeros(Fs).cuda()
    d = L.EmbDedupDesc(); d.kind = L.OP_EMB_DEDUP; d.B, d.Fs = B, Fs
    d.idx, d.dout, d.leader, d.gsum, d.sumsq_partial = idx.data_ptr(), dout.data_ptr(), leader.data_ptr(), gsum.data_ptr(), part.data_ptr()
    with torch.cuda.stream(st):
        print(label, "%.1f us" % (time_desc(lib, L, sp, d) * 1e3))
run(T, "criteo tables")
run([10**7]*26, "all unique")
run([4]*26, "all n=4")
