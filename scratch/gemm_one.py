import sys, ctypes as C; sys.path.insert(0,'/root/repo')
import torch
from nasrec_amd import _lib as L
lib = L.load(); st = torch.cuda.Stream(); sp = st.cuda_stream
M,N,K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A = torch.randn(M*K+1024, device='cuda'); Bm = torch.randn(N*K+1024, device='cuda'); Cc = torch.zeros(M*N+1024, device='cuda')
d = L.GemmDesc(); d.kind=L.OP_GEMM; d.amode,d.bmode,d.cmode=L.AM_KC,L.AM_KC,L.CM_PLAIN; d.nseg=1; d.zmode=0; d.dims_in_use=-1; d.splitk=1
s=d.seg[0]; s.A=A.data_ptr(); s.B=Bm.data_ptr(); s.C=Cc.data_ptr(); s.M=M; s.N=N; s.K=K; s.lda=K; s.ldb=K; s.ldc=N; s.Mvalid=M
with torch.cuda.stream(st):
    for _ in range(5): L.check(lib.nasrec_launch(sp, C.addressof(d)))
st.synchronize()
