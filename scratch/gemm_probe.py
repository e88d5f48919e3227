import sys, ctypes as C; sys.path.insert(0,'/root/repo')
import torch
from nasrec_amd import _lib as L
sys.path.insert(0,'/root/repo'); from bench import time_desc
lib = L.load(); st = torch.cuda.Stream(); sp = st.cuda_stream
def mk(am,bm,cm,M,N,K,lda,ldb,ldc,S=1,ones=0):
    A = torch.randn(max(M,K)*max(lda,1)+1024, device='cuda'); Bm = torch.randn(max(N,K)*max(ldb,1)+1024, device='cuda'); Cc = torch.zeros(M*max(ldc,N)+1024, device='cuda')
    d = L.GemmDesc(); d.kind=L.OP_GEMM; d.amode,d.bmode,d.cmode=am,bm,cm; d.nseg=1; d.zmode=1; d.dims_in_use=-1; d.splitk=S
    ws = torch.zeros(S*M*(N+1)+16, device='cuda'); cnt = torch.zeros(4096, dtype=torch.int32, device='cuda'); rs = torch.zeros(M+8, device='cuda')
    d.workspace=ws.data_ptr(); d.counters=cnt.data_ptr(); d.rowsum_out=rs.data_ptr()
    s=d.seg[0]; s.A=A.data_ptr(); s.B=Bm.data_ptr(); s.C=Cc.data_ptr(); s.M=M; s.N=N+ones; s.K=K; s.lda=lda; s.ldb=ldb; s.ldc=ldc; s.Mvalid=M; s.ones_col=ones
    return d,(A,Bm,Cc,ws,cnt,rs)
with torch.cuda.stream(st):
    for (M,N,K) in [(768,780,256),(768,1565,256),(16,13,256),(128,813,256)]:
        for S in (1,4):
            for ones in (0,1):
                d,k = mk(L.AM_RC,L.AM_RC,L.CM_PLAIN,M,N,K,M,N,N,S,ones)
                print("dW  M=%4d N=%4d K=%d S=%d ones=%d: %.1f us" % (M,N,K,S,ones,time_desc(lib,L,sp,d)*1e3))
    for (M,N,K) in [(256,768,1565),(256,16,1581),(256,128,813),(256,768,780)]:
        for S in (1,2,4,8):
            d,k = mk(L.AM_KC,L.AM_KC,L.CM_PLAIN,M,N,K,K,K,N,S,0)
            print("fwd M=%4d N=%4d K=%d S=%d: %.1f us" % (M,N,K,S,time_desc(lib,L,sp,d)*1e3))
    for (M,N,K) in [(48,194,4096),(64,26,4096)]:
        for S in (1,4,8,16):
            d,k = mk(L.AM_TOKK,L.AM_TOKK,L.CM_PLAIN,M,N,K,M*16,N*16,N,S,0)
            print("tokdW M=%4d N=%4d K=%d S=%d: %.1f us" % (M,N,K,S,time_desc(lib,L,sp,d)*1e3))
