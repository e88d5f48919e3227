"""The row-sparse embedding backward in two halves (csrc/dedup_bodies.h: NASREC_OP_DEDUP_IDS any time after the ids are known,
NASREC_OP_OPT_REDUCE2 behind the backward), through the C-ABI, against
  * the one-launch kernels (NASREC_OP_EMB_DEDUP: one workgroup per field at B <= 256, chunk + merge above): same leaders, the summed rows
    BIT for bit at every batch size (same summation order: sub-runs ascending, then chunk order),
  * a sequential fp32 loop in that order, and the fp64 dense gradient (index_add_) / its norm,
and the whole tail OPT_REDUCE2 + OPT_APPLY against clip_grad_norm_ + torch.optim.Adagrad on dense gradients (the reference step,
nasrec/utils/train_utils.py:283-286), contiguous rows and the receive-buffer layout of a data-parallel step's all-gather."""
import ctypes as C

import pytest
import torch

from nasrec_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return L.load()


def dev(t):
    return t.cuda()


def launch(lib, d):
    L.check(lib.nasrec_launch(torch.cuda.current_stream().cuda_stream, C.addressof(d)))


def cap_of(B):
    c = 256
    while c < B:
        c *= 2
    return c


def two_halves(lib, idx_d, rows_d, B, Fs, rank=None, nsq=0, g=None):
    """-> (leader [B,Fs] int32, partial sums, descriptors): DEDUP_IDS then OPT_REDUCE2 in place over rows_d"""
    cap = cap_of(B)
    bufs = dict(leader=dev(torch.full((B * Fs,), 9, dtype=torch.int32)), order=dev(torch.zeros(Fs * cap, dtype=torch.int32)),
                lists=dev(torch.zeros(Fs * cap, dtype=torch.int32)), counts=dev(torch.zeros(Fs * 2, dtype=torch.int32)),
                heads=dev(torch.zeros(Fs * cap, dtype=torch.int32)))
    ids = L.DedupIdsDesc()
    ids.kind, ids.B, ids.Fs, ids.cap = L.OP_DEDUP_IDS, B, Fs, cap
    ids.idx, ids.leader, ids.order, ids.lists, ids.counts = idx_d.data_ptr(), bufs["leader"].data_ptr(), bufs["order"].data_ptr(), bufs["lists"].data_ptr(), bufs["counts"].data_ptr()
    ids.heads = bufs["heads"].data_ptr() if B > 256 else None
    launch(lib, ids)
    rb = max(1, min(512, (B * Fs * 4 + 255) // 256))
    part = dev(torch.full((Fs + rb,), 7.0))
    r2 = L.OptReduce2Desc()
    r2.kind, r2.B, r2.Fs, r2.cap, r2.row_blocks = L.OP_OPT_REDUCE2, B, Fs, cap, rb
    if rank:
        r2.rank_B, r2.rank_stride = rank
    r2.rows, r2.leader, r2.order, r2.lists, r2.counts, r2.heads = rows_d.data_ptr(), ids.leader, ids.order, ids.lists, ids.counts, ids.heads
    r2.sumsq_partial = part.data_ptr()
    pd = None
    if nsq:
        pd = dev(torch.zeros(nsq))
        r2.sumsq.kind, r2.sumsq.nblocks, r2.sumsq.n, r2.sumsq.x, r2.sumsq.partial = L.OP_SUMSQ, nsq, g.numel(), g.data_ptr(), pd.data_ptr()
    launch(lib, r2)
    return bufs, part, pd, (ids, r2)


def reference_sum(idx, dout, f):
    """fp32 rows summed in the engine's order: inside a 256-sample chunk ascending, chunk sums into the first chunk's in chunk order"""
    B = idx.shape[0]
    chunk_first, chunk_acc, order = {}, {}, {}
    for b in range(B):
        key = (int(idx[b, f]), b >> 8)
        if key in chunk_first:
            chunk_acc[key] = chunk_acc[key] + dout[b, f]
        else:
            chunk_first[key], chunk_acc[key] = b, dout[b, f].clone()
            order.setdefault(key[0], []).append(key)
    first, acc = {}, {}
    for i, keys in order.items():
        first[i] = chunk_first[keys[0]]
        a = chunk_acc[keys[0]].clone()
        for k in keys[1:]:
            a = a + chunk_acc[k]
        acc[i] = a
    return first, acc


@pytest.mark.parametrize("B", [1, 37, 255, 256, 257, 700, 1024, 1500, 2048])
def test_two_halves_equal_the_one_launch_dedup_bit_for_bit(lib, B):
    torch.manual_seed(100 + B)
    rows = [4, 1, 7, 50, 3000, 10 ** 6]
    Fs = len(rows)
    idx = torch.stack([torch.randint(0, n, (B,)) for n in rows], 1)
    dout = torch.randn(B, Fs, 16)
    gi = dev(idx)
    # the one-launch kernels
    leader0, gsum0 = dev(torch.zeros(B * Fs, dtype=torch.int32)), dev(torch.zeros(B * Fs * 16))
    part0 = dev(torch.zeros(Fs * ((B + 255) // 256)))
    dd = L.EmbDedupDesc()
    dd.kind, dd.B, dd.Fs = L.OP_EMB_DEDUP, B, Fs
    gd0 = dev(dout)
    dd.idx, dd.dout, dd.leader, dd.gsum, dd.sumsq_partial = gi.data_ptr(), gd0.data_ptr(), leader0.data_ptr(), gsum0.data_ptr(), part0.data_ptr()
    launch(lib, dd)
    # the two halves, in place
    rows_d = dev(dout.clone())
    bufs, part, _, _ = two_halves(lib, gi, rows_d, B, Fs)
    torch.cuda.synchronize()
    lead0 = leader0.cpu().view(B, Fs)
    lead = bufs["leader"].cpu().view(B, Fs)
    assert int(lead.max()) <= 2 and int(lead.min()) >= 0
    assert torch.equal((lead > 0).int(), lead0), "leaders differ from the one-launch kernels"
    got, want = rows_d.cpu().view(B, Fs, 16), gsum0.cpu().view(B, Fs, 16)
    sel = lead0.bool()
    assert torch.equal(got[sel], want[sel]), "summed rows are not the one-launch kernels' bits"
    total = 0.0
    for f in range(Fs):
        first, acc = reference_sum(idx, dout, f)
        counts = torch.bincount(idx[:, f], minlength=1)
        for i, b in first.items():
            assert int(lead[b, f]) == (2 if int(counts[i]) > 1 else 1), (f, i)
            assert torch.equal(got[b, f], acc[i]), "field %d row %d: not the chunk-ordered fp32 sum" % (f, i)
            total += float(acc[i].double().pow(2).sum())
    assert abs(float(part.double().sum()) - total) <= 2e-6 * total
    assert abs(float(part0.double().sum()) - total) <= 1e-5 * total
    # bit-reproducible (lists, sums and partials)
    rows2 = dev(dout.clone())
    bufs2, part2, _, _ = two_halves(lib, gi, rows2, B, Fs)
    torch.cuda.synchronize()
    assert torch.equal(part, part2) and torch.equal(rows2.view(B, Fs, 16)[sel.cuda()], rows_d.view(B, Fs, 16)[sel.cuda()])
    for k in ("leader", "counts"):
        assert torch.equal(bufs[k], bufs2[k]), k
    cap = cap_of(B)
    for f in range(Fs):
        if B <= 256:  # compact list of the runs with duplicates (entries behind a field's count are stale memory)
            nA = int(bufs["counts"][2 * f])
            assert int(bufs["counts"][2 * f + 1]) == 0
            assert torch.equal(bufs["lists"][f * cap:f * cap + nA], bufs2["lists"][f * cap:f * cap + nA])
            entries = [(int(bufs["lists"][f * cap + k]) & 0xffff, (int(bufs["lists"][f * cap + k]) >> 16) & 0x7fff) for k in range(nA)]
        else:  # one entry and one next-sub-run link per sample
            nb = (B + 255) // 256 * 256
            assert torch.equal(bufs["lists"][f * cap:f * cap + nb], bufs2["lists"][f * cap:f * cap + nb])
            assert torch.equal(bufs["heads"][f * cap:f * cap + nb], bufs2["heads"][f * cap:f * cap + nb])
            ens = (bufs["lists"][f * cap:f * cap + nb].cpu().long() & 0xffffffff).tolist()
            entries = [((b & ~255) + (en & 0xff), 1 + ((en >> 8) & 0xff)) for b, en in enumerate(ens) if en & 0x10000]
            nxt = bufs["heads"][f * cap:f * cap + nb].cpu().tolist()
            col = idx[:, f].tolist()
            for b, en in enumerate(ens):
                if en & 0x20000:  # leader of a run that spans chunks: the chain visits the first sample of every later chunk's part
                    chain, h = [], nxt[b]
                    while h >= 0:
                        chain.append(h)
                        h = nxt[h]
                    same = [j for j in range(B) if col[j] == col[b]]
                    assert same[0] == b
                    firsts = {}
                    for j in same:
                        firsts.setdefault(j >> 8, j)
                    assert [b] + chain == [firsts[c] for c in sorted(firsts)], (f, b)
        for s0, ln in entries:  # the sub-runs' samples
            assert torch.equal(bufs["order"][f * cap + s0:f * cap + s0 + ln], bufs2["order"][f * cap + s0:f * cap + s0 + ln])
            o = bufs["order"][f * cap + s0:f * cap + s0 + ln].cpu().long()
            assert ln >= 2 and bool((o[1:] > o[:-1]).all()) and len(set(idx[o, f].tolist())) == 1 and len(set((o >> 8).tolist())) == 1


def _dd_hash(i):
    return ((i * 2654435761) & 0xffffffff) >> 20  # csrc/dedup_bodies.h dd_hash


@pytest.mark.parametrize("case", ["one_id", "all_distinct", "one_hash_bucket", "pairs_across_chunks"])
def test_id_half_on_adversarial_ids(lib, case):
    """the id half above 256 samples goes through an LDS hash table of the other chunks' ids (open addressing): one id everywhere (a
    single run of 8 sub-runs of 256), no duplicates at all, every id in the same bucket (a probe chain as long as the table is full),
    and every id exactly twice, 1024 samples apart"""
    B, Fs = 2048, 2
    g = torch.Generator().manual_seed(11)
    if case == "one_id":
        col = torch.full((B,), 12345, dtype=torch.int64)
    elif case == "all_distinct":
        col = torch.randperm(10 ** 6, generator=g)[:B]
    elif case == "one_hash_bucket":
        pool, i = [], 1
        while len(pool) < 600:  # 600 distinct ids with dd_hash == 77, each about 3.4 times
            if _dd_hash(i) == 77:
                pool.append(i)
            i += 1
        col = torch.tensor(pool)[torch.randint(0, len(pool), (B,), generator=g)]
    else:
        half = torch.randperm(10 ** 6, generator=g)[:B // 2]
        col = torch.cat([half, half])
    idx = torch.stack([col, torch.randint(0, 50, (B,), generator=g)], 1)
    dout = torch.randn(B, Fs, 16, generator=g)
    gi, rows_d = dev(idx), dev(dout.clone())
    bufs, part, _, _ = two_halves(lib, gi, rows_d, B, Fs)
    torch.cuda.synchronize()
    lead, got = bufs["leader"].cpu().view(B, Fs), rows_d.cpu().view(B, Fs, 16)
    total = 0.0
    for f in range(Fs):
        first, acc = reference_sum(idx, dout, f)
        assert int((lead[:, f] > 0).sum()) == len(first)
        counts = torch.bincount(idx[:, f])
        for i, b in first.items():
            assert int(lead[b, f]) == (2 if int(counts[i]) > 1 else 1), (f, i)
            assert torch.equal(got[b, f], acc[i]), (case, f, i)
            total += float(acc[i].double().pow(2).sum())
    assert abs(float(part.double().sum()) - total) <= 2e-6 * total


@pytest.mark.parametrize("B,ranks", [(256, 1), (512, 2), (2048, 8), (300, 3)])
def test_two_halves_tail_equals_the_dense_reference_step(lib, B, ranks):
    """OPT_REDUCE2 + OPT_APPLY == embedding_dense_backward + clip_grad_norm_ + torch.optim.Adagrad (dense, fp64), two steps, with the
    rows in the receive buffer of a `ranks`-way all-gather (something else behind every rank's rows) and contiguous"""
    torch.manual_seed(4 + B)
    rows = [4, 50, 3000]
    Fs, n = 3, 50001
    tables = [torch.randn(r, 16) for r in rows]
    idx = torch.stack([torch.randint(0, r, (B,)) for r in rows], 1)
    dout = torch.randn(B, Fs, 16) * 0.1
    g = torch.randn(n) * 0.05
    p0 = torch.randn(n)
    lr, eps, clip = 0.16, 1e-2, 0.05
    for layout in ("contiguous", "gathered"):
        ref_t = [torch.nn.Parameter(t.double().clone()) for t in tables]
        ref_p = torch.nn.Parameter(p0.double().clone())
        opt = torch.optim.Adagrad(ref_t + [ref_p], lr=lr, eps=eps)
        gt, gs = [dev(t.clone()) for t in tables], [dev(torch.zeros_like(t)) for t in tables]
        gp, gst, gg = dev(p0.clone()), dev(torch.zeros(n)), dev(g)
        gi = dev(idx)
        lr_d, coef = dev(torch.tensor([lr])), dev(torch.zeros(2))
        Bl = B // ranks
        rank = None
        if layout == "gathered":
            stride = Bl * Fs * 16 + 1000
            rank = (Bl, stride)
        for step in range(2):
            for f in range(Fs):
                ref_t[f].grad = torch.zeros_like(ref_t[f]).index_add_(0, idx[:, f], dout[:, f].double())
            ref_p.grad = g.double().clone()
            total = torch.nn.utils.clip_grad_norm_(ref_t + [ref_p], clip)
            opt.step()
            if rank:
                buf = torch.full((ranks * rank[1],), 123.0)
                buf.view(ranks, rank[1])[:, :Bl * Fs * 16] = dout.view(ranks, Bl * Fs * 16)
                rows_d = dev(buf)
            else:
                rows_d = dev(dout.clone())
            nsq = 19
            bufs, part, pd, (ids, r2) = two_halves(lib, gi, rows_d, B, Fs, rank=rank, nsq=nsq, g=gg)
            app = L.OptApplyDesc()
            app.kind, app.dense_blocks = L.OP_OPT_APPLY, min(1024, (n + 255) // 256)
            app.clip.kind, app.clip.n_a, app.clip.n_b, app.clip.max_norm = L.OP_CLIP_COEF, nsq, part.numel(), clip
            app.clip.partial_a, app.clip.partial_b, app.clip.out = pd.data_ptr(), part.data_ptr(), coef.data_ptr()
            app.dense.kind, app.dense.eps, app.dense.n = L.OP_ADAGRAD_DENSE, eps, n
            app.dense.p, app.dense.g, app.dense.state, app.dense.lr, app.dense.coef = gp.data_ptr(), gg.data_ptr(), gst.data_ptr(), lr_d.data_ptr(), coef.data_ptr()
            ar = app.rows
            ar.kind, ar.B, ar.Fs, ar.eps = L.OP_ADAGRAD_ROWS, B, Fs, eps
            ar.idx, ar.leader, ar.gsum, ar.lr, ar.coef = gi.data_ptr(), bufs["leader"].data_ptr(), rows_d.data_ptr(), lr_d.data_ptr(), coef.data_ptr()
            if rank:
                ar.rank_B, ar.rank_stride = rank
            for f in range(Fs):
                ar.table[f], ar.state[f], ar.rows[f] = gt[f].data_ptr(), gs[f].data_ptr(), rows[f]
            launch(lib, app)
            torch.cuda.synchronize()
            assert abs(float(coef[1]) - float(total)) <= 1e-5 * float(total), layout
            if rank:  # what rides behind the rows is not touched
                assert torch.equal(rows_d.cpu().view(ranks, rank[1])[:, Bl * Fs * 16:], torch.full((ranks, 1000), 123.0))
            for f in range(Fs):
                assert float((gt[f].cpu().double() - ref_t[f].data).abs().max()) <= 1e-5, (layout, f)
            assert float((gp.cpu().double() - ref_p.data).abs().max()) <= 1e-5, layout


def test_the_id_half_rides_on_the_staging_launch(lib):
    """NASREC_OP_STAGE_INPUTS with stage.dedup_ids set: the same leaders / order / lists as the stand-alone launch on the staged ids"""
    torch.manual_seed(3)
    B, Fd, Fs = 200, 5, 4
    rows = [4, 50, 3000, 2]
    tabs = [dev(torch.randn(r, 16)) for r in rows]
    idx = dev(torch.stack([torch.randint(0, r, (B,)) for r in rows], 1))
    int_src, y_src = dev(torch.randn(B, Fd)), dev(torch.rand(B))
    int_dst, cat_dst, y_dst = dev(torch.zeros(B, Fd)), dev(torch.zeros(B, Fs, dtype=torch.int64)), dev(torch.zeros(B))
    out = dev(torch.zeros(B, Fs, 16))
    oob = dev(torch.zeros(1, dtype=torch.int32))

    def id_bufs():
        return dict(leader=dev(torch.full((B * Fs,), 9, dtype=torch.int32)), order=dev(torch.zeros(Fs * 256, dtype=torch.int32)),
                    lists=dev(torch.zeros(Fs * 256, dtype=torch.int32)), counts=dev(torch.zeros(Fs * 2, dtype=torch.int32)))
    a, b = id_bufs(), id_bufs()
    st = L.StageDesc()
    st.kind, st.B, st.Fd, st.Fs = L.OP_STAGE_INPUTS, B, Fd, Fs
    st.int_src, st.int_dst, st.cat_src, st.cat_dst, st.y_src, st.y_dst = int_src.data_ptr(), int_dst.data_ptr(), idx.data_ptr(), cat_dst.data_ptr(), y_src.data_ptr(), y_dst.data_ptr()
    st.gather.B, st.gather.Fs, st.gather.out, st.gather.oob = B, Fs, out.data_ptr(), oob.data_ptr()
    for f in range(Fs):
        st.gather.table[f], st.gather.rows[f] = tabs[f].data_ptr(), rows[f]
    st.dedup_ids.cap = 256
    st.dedup_ids.leader, st.dedup_ids.order, st.dedup_ids.lists, st.dedup_ids.counts = a["leader"].data_ptr(), a["order"].data_ptr(), a["lists"].data_ptr(), a["counts"].data_ptr()
    launch(lib, st)
    ids = L.DedupIdsDesc()
    ids.kind, ids.B, ids.Fs, ids.cap, ids.idx = L.OP_DEDUP_IDS, B, Fs, 256, idx.data_ptr()
    ids.leader, ids.order, ids.lists, ids.counts = b["leader"].data_ptr(), b["order"].data_ptr(), b["lists"].data_ptr(), b["counts"].data_ptr()
    launch(lib, ids)
    torch.cuda.synchronize()
    assert torch.equal(cat_dst, idx) and torch.equal(int_dst, int_src) and torch.equal(y_dst, y_src) and int(oob.item()) == 0
    assert torch.equal(out, torch.stack([tabs[f][idx[:, f]] for f in range(Fs)], 1))
    assert torch.equal(a["leader"], b["leader"]) and torch.equal(a["counts"], b["counts"])
    for f in range(Fs):
        nA = int(a["counts"][2 * f])
        assert int(a["counts"][2 * f + 1]) == 0  # one chunk: no run spans chunks
        assert torch.equal(a["lists"][f * 256:f * 256 + nA], b["lists"][f * 256:f * 256 + nA])
        used, want_lead = 0, {}
        for bb in range(B):
            want_lead.setdefault(int(idx[bb, f]), bb)
        for k in range(nA):  # every run with duplicates: leader first, then its duplicates ascending
            en = int(a["lists"][f * 256 + k]) & 0xffffffff
            s0, ln = en & 0xffff, (en >> 16) & 0x7fff
            assert en >> 31 == 1 and s0 == used
            used += ln
            o = a["order"][f * 256 + s0:f * 256 + s0 + ln]
            assert torch.equal(o, b["order"][f * 256 + s0:f * 256 + s0 + ln])
            o = o.cpu().long()
            ids_of = idx[:, f].cpu()[o]
            assert bool((o[1:] > o[:-1]).all()) and len(set(ids_of.tolist())) == 1 and want_lead[int(ids_of[0])] == int(o[0])
            assert ln == int((idx[:, f] == ids_of[0]).sum())
