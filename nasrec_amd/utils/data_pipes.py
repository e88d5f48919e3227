"""Input pipeline of the training harness: sharded TSV readers + column transforms with the reference's contract
(nasrec/utils/data_pipes.py, nasrec/torchrec/{criteo,avazu,kdd,utils}.py), pinned by tests/golden/datapipes.npz which was
produced by the real reference readers.

Row format (tab separated, parsed with the `csv` module like the reference): label, Fd integer columns, Fs hexadecimal
categorical columns.  Casting: label / integers via int() with 0 for anything unparsable (torchrec/utils.py safe_cast);
an empty categorical field is a missing id.  Transforms (data_pipes.py:137-175):
    int_x = log(max(0, x) + 1)   float32  (Avazu: zeros — it has no dense features, data_pipes.py:178-183)
    cat_x = fmod(int(v, 16), n_f - 1) + 1, missing -> fmod(-1, n_f - 1) + 1 = 0      int64
    y     = label as float32 [B, 1]
Batches are cut per shard (the last one of a shard may be short); shards are interleaved round-robin, which is the order
a DataLoader with one worker per shard delivers them (ParallelReadConcat, torchrec/utils.py:225-302).

Two interchangeable readers produce those batches: the pure-Python one (csv module, ~26 k rows/s per process — the
reference's own speed) and the native one (`nasrec_tsv_parse` in the engine library, one thread per shard with a bounded
prefetch queue; lines on which only Python's parsers define the result are handed back and parsed here), selected by
NASREC_TSV_READER = native (default) | python.  Both are tested against the same golden batches.

`--root_dir synthetic[:key=value,...]` (extension; the datasets are not redistributable) yields Criteo/Avazu/KDD-shaped random
batches instead: keys steps (batches per epoch), test_steps, seed."""
import csv
import ctypes as C
import glob
import io
import os
import queue
import threading
from typing import Iterator, List, Tuple

import numpy as np
import torch

from .config import NUM_EMBEDDINGS_AVAZU, NUM_EMBEDDINGS_CRITEO, NUM_EMBEDDINGS_KDD

Batch = Tuple[torch.Tensor, torch.Tensor, torch.Tensor]


class DatasetSpec:
    def __init__(self, name, num_dense, num_sparse, tables, dense_is_zero=False):
        self.name, self.Fd, self.Fs, self.tables, self.dense_is_zero = name, num_dense, num_sparse, list(tables), dense_is_zero


SPECS = {
    "criteo-kaggle": DatasetSpec("criteo-kaggle", 13, 26, NUM_EMBEDDINGS_CRITEO),
    "avazu": DatasetSpec("avazu", 1, 23, NUM_EMBEDDINGS_AVAZU, dense_is_zero=True),
    "kdd": DatasetSpec("kdd", 3, 10, NUM_EMBEDDINGS_KDD),
}


def _to_int(text, base=10):
    try:
        return int(text, base)
    except ValueError:
        return 0


def rows_to_batch(rows: List[List[str]], spec: DatasetSpec) -> Batch:
    """rows: split TSV lines of one batch -> (int_x f32 [B,Fd], cat_x i64 [B,Fs], y f32 [B,1])"""
    width = 1 + spec.Fd + spec.Fs
    for r in rows:
        if len(r) != width:
            raise ValueError("expected %d tab-separated columns for %s, got %d" % (width, spec.name, len(r)))
    y = torch.tensor([[_to_int(r[0])] for r in rows], dtype=torch.int64).float()
    dense = torch.tensor([[_to_int(v) for v in r[1:1 + spec.Fd]] for r in rows], dtype=torch.int64).view(len(rows), spec.Fd)
    if spec.dense_is_zero:
        int_x = torch.zeros(len(rows), spec.Fd, dtype=torch.float32)
    else:
        int_x = torch.log(torch.clamp_min(dense, 0) + 1)  # integer tensor -> float32 log, as the reference computes it
    ids = torch.tensor([[int(v, 16) if v else -1 for v in r[1 + spec.Fd:]] for r in rows], dtype=torch.int64).view(len(rows), spec.Fs)
    mod = torch.tensor(spec.tables, dtype=torch.int64) - 1
    cat_x = torch.fmod(ids, mod) + 1
    return int_x, cat_x, y


class TsvShard:
    """one TSV file, cut into batches of `batch_size` rows (re-iterable)"""

    def __init__(self, path: str, spec: DatasetSpec, batch_size: int):
        self.path, self.spec, self.batch_size = path, spec, batch_size

    def __iter__(self) -> Iterator[Batch]:
        with open(self.path, "r", newline="") as f:
            rows = []
            for row in csv.reader(f, delimiter="\t"):
                rows.append(row)
                if len(rows) == self.batch_size:
                    yield rows_to_batch(rows, self.spec)
                    rows = []
            if rows:
                yield rows_to_batch(rows, self.spec)


class NativeTsvShard(TsvShard):
    """the same batches as TsvShard, parsed by nasrec_tsv_parse (C-ABI) in chunks of `chunk_bytes`.

    The rows of one parser call (~16 k) become ONE set of tensors — label, the log transform, the ids: three tensor operations per
    16 k rows — and the batches are views of them: assembling every 256-row batch with its own tensor calls held the interpreter lock
    for ~80 us per batch and capped eight reader threads at 0.4 M rows/s, below what the engine consumes at batch 256.
    `iter_super(pin=True)` yields those row blocks in pinned host memory for the device staging of RoundRobinLoader."""

    chunk_bytes = 4 << 20

    def _assemble(self, label, dense, cat, pin=False):
        spec = self.spec
        m = len(label)
        if not pin:
            y = torch.from_numpy(label).float().view(-1, 1)
            if spec.dense_is_zero:
                int_x = torch.zeros(m, spec.Fd, dtype=torch.float32)
            else:
                int_x = torch.log(torch.clamp_min(torch.from_numpy(dense), 0) + 1)
            return int_x, torch.from_numpy(cat), y
        y = torch.empty(m, 1, dtype=torch.float32, pin_memory=True)
        y.copy_(torch.from_numpy(label).view(-1, 1))
        int_x = torch.empty(m, spec.Fd, dtype=torch.float32, pin_memory=True)
        if spec.dense_is_zero:
            int_x.zero_()
        else:
            torch.log(torch.clamp_min(torch.from_numpy(dense), 0) + 1, out=int_x)
        c = torch.empty(m, spec.Fs, dtype=torch.int64, pin_memory=True)
        c.copy_(torch.from_numpy(cat))
        return int_x, c, y

    def __iter__(self) -> Iterator[Batch]:
        bs = self.batch_size
        for int_x, cat_x, y in self.iter_super():
            for i in range(0, len(y), bs):
                yield int_x[i:i + bs], cat_x[i:i + bs], y[i:i + bs]

    def iter_super(self, pin=False) -> Iterator[Batch]:
        """row blocks (int_x, cat_x, y) whose length is a multiple of the batch size (the shard's last block may end in a short batch)"""
        from .. import _lib as L
        lib = L.load()
        spec, bs = self.spec, self.batch_size
        tables = np.asarray(spec.tables, dtype=np.int64)
        cap = bs * max(1, 16384 // bs)  # rows per parser call
        consumed, status = C.c_int64(), C.c_int32()

        def fresh(carry=None):
            # rows are parsed straight into these arrays; the < bs rows left over after a block move to the front of the next set
            arrs = [np.empty(cap + 2 * bs, np.int64), np.empty((cap + 2 * bs, spec.Fd), np.int64), np.empty((cap + 2 * bs, spec.Fs), np.int64)]
            k = 0
            if carry is not None:
                k = len(carry[0])
                for a, c in zip(arrs, carry):
                    a[:k] = c
            return arrs, k

        arrs, fill = fresh()

        def emit(final=False):
            nonlocal arrs, fill
            nb = fill // bs
            m = fill if final else nb * bs
            if m:
                yield self._assemble(arrs[0][:m], arrs[1][:m], arrs[2][:m], pin)
            rest = fill - m
            if m or final:
                arrs, fill = fresh([a[m:m + rest] for a in arrs] if rest else None)

        with open(self.path, "rb") as f:
            tail = b""
            while True:
                chunk = f.read(self.chunk_bytes)
                eof = not chunk
                buf = tail + chunk
                if eof and buf and not buf.endswith(b"\n"):
                    buf += b"\n"
                pos = 0
                base = C.cast(C.c_char_p(buf), C.c_void_p).value or 0  # buf stays referenced for the duration of the calls
                while pos < len(buf):
                    room = cap + bs - fill
                    n = lib.nasrec_tsv_parse(base + pos, len(buf) - pos, spec.Fd, spec.Fs, tables.ctypes.data, room,
                                             arrs[0][fill:].ctypes.data, arrs[1][fill:].ctypes.data, arrs[2][fill:].ctypes.data,
                                             C.byref(consumed), C.byref(status))
                    fill += n
                    pos += consumed.value
                    if status.value == L.TSV_NEEDS_PYTHON or status.value == L.TSV_BAD_COLUMNS:
                        end = buf.find(b"\n", pos)
                        line = buf[pos:end].decode("utf-8")
                        rows = list(csv.reader(io.StringIO(line, newline=""), delimiter="\t")) or [[]]
                        _, bc, by = rows_to_batch(rows, spec)  # raises on a wrong column count, like the Python reader
                        if fill + len(rows) > cap + 2 * bs:
                            yield from emit()
                        k = len(rows)  # back to the raw integer form (int_x is recomputed identically on assembly)
                        arrs[0][fill:fill + k] = by.view(-1).numpy().astype(np.int64)
                        arrs[1][fill:fill + k] = np.asarray([[_to_int(v) for v in r[1:1 + spec.Fd]] for r in rows], np.int64).reshape(k, spec.Fd)
                        arrs[2][fill:fill + k] = bc.numpy()
                        fill += k
                        pos = end + 1
                    elif n == 0 and consumed.value == 0:
                        break  # only an incomplete line is left in this chunk
                    if fill >= cap:
                        yield from emit()
                tail = buf[pos:]
                if eof:
                    break
        yield from emit(final=True)


class _Prefetch:
    """runs an iterable in a background thread with a bounded queue (the C parser releases the GIL, so one thread per
    shard parses in parallel).  `close()` (also on garbage collection, and from RoundRobinLoader when its iterator is abandoned —
    `next(iter(loader))`, `break` on max_train_steps) stops the thread and drops what it had queued: with device staging a queue
    item is a ~16 k-row block on the GPU plus its pinned host copy, and an abandoned producer blocked in `put` would hold
    several of them, and an open file, for the life of the process."""

    _END = object()

    def __init__(self, source, depth=4):
        self.q = queue.Queue(maxsize=depth)
        self.err = None
        self._stop = threading.Event()
        self.t = threading.Thread(target=self._run, args=(source,), daemon=True)
        self.t.start()

    def _put(self, item) -> bool:
        while not self._stop.is_set():
            try:
                self.q.put(item, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def _run(self, source):
        it = iter(source)
        try:
            for item in it:
                if not self._put(item):
                    break
                del item
        except BaseException as e:  # surfaced in the consumer
            self.err = e
        finally:
            close = getattr(it, "close", None)
            if close is not None:  # a generator: run its clean-up (file handles) now, in this thread
                try:
                    close()
                except Exception:  # noqa: BLE001
                    pass
        self._put(self._END)

    def close(self):
        self._stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        if self.t is not threading.current_thread():
            self.t.join(timeout=2.0)
        try:  # (an item the producer slipped in between the drain and its exit)
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __iter__(self):
        return self

    def __next__(self):
        if self._stop.is_set():
            raise StopIteration
        item = self.q.get()
        if item is self._END:
            if self.err is not None:
                raise self.err
            raise StopIteration
        return item


class _DeviceBlocks:
    """a shard's row blocks staged to the GPU: parsed into pinned host memory by the shard's thread, copied with non-blocking H2D
    copies on a side stream (the reference does three blocking `.to(gpu)` per batch on the compute stream, train_utils.py:257-259),
    handed out as device batches once the copy's event has been waited for on the consumer's stream"""

    def __init__(self, shard, device, stream):
        self.shard, self.device, self.stream = shard, device, stream

    def __iter__(self):
        for host in self.shard.iter_super(pin=True):
            with torch.cuda.stream(self.stream):
                dev = tuple(t.to(self.device, non_blocking=True) for t in host)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            yield dev, ev, host  # (the pinned block stays referenced until the consumer drops the item)


class RoundRobinLoader:
    """interleaves the batches of several shards in the order a one-worker-per-shard DataLoader delivers them; with
    `prefetch` every shard is read by its own background thread; with `device` (native shards) the batches arrive ON the GPU,
    staged block-wise through pinned memory on a second stream"""

    def __init__(self, pipes, prefetch=False, device=None):
        self.pipes, self.prefetch = list(pipes), prefetch
        self.device = torch.device("cuda", device) if isinstance(device, int) else (torch.device(device) if device is not None else None)
        if self.device is not None and (self.device.type != "cuda" or not all(isinstance(p, NativeTsvShard) for p in self.pipes)):
            self.device = None
        self._stream = None

    def _device_batches(self, blocks, bs):
        cur = torch.cuda.current_stream(self.device)
        for dev, ev, host in blocks:
            cur.wait_event(ev)
            for t in dev:
                t.record_stream(cur)
            int_x, cat_x, y = dev
            for i in range(0, len(y), bs):
                yield int_x[i:i + bs], cat_x[i:i + bs], y[i:i + bs]

    def peek(self) -> Batch:
        """the first batch (what `next(iter(loader))` returns — the reference peeks one batch per epoch and per warm-up,
        train_utils.py:224-225,392-433) read on the host path: no reader threads, no device blocks, nothing left behind"""
        it = iter(self.pipes[0])
        try:
            return next(it)
        finally:
            close = getattr(it, "close", None)
            if close is not None:
                close()

    def __iter__(self) -> Iterator[Batch]:
        threads = []
        if self.device is not None:
            if self._stream is None:
                self._stream = torch.cuda.Stream(self.device)
            threads = [_Prefetch(_DeviceBlocks(p, self.device, self._stream), depth=3) for p in self.pipes]
            live = [iter(self._device_batches(t, p.batch_size)) for t, p in zip(threads, self.pipes)]
        elif self.prefetch:
            live = threads = [_Prefetch(p) for p in self.pipes]
        else:
            live = [iter(p) for p in self.pipes]
        try:
            while live:
                nxt = []
                for it in live:
                    try:
                        yield next(it)
                        nxt.append(it)
                    except StopIteration:
                        pass
                live = nxt
        finally:  # also reached through GeneratorExit when the consumer abandons the iterator
            live = None
            for t in threads:
                t.close()


class SyntheticPipe:
    """SURVEY §8d synthetic inputs: int_x = log(U_int[0,1000) + 1) (Avazu zeros), ids uniform per table, y ~ Bernoulli(0.25)"""

    def __init__(self, spec: DatasetSpec, batch_size: int, steps: int, seed: int):
        self.spec, self.batch_size, self.steps, self.seed = spec, batch_size, steps, seed

    def __iter__(self) -> Iterator[Batch]:
        g = torch.Generator().manual_seed(self.seed)
        B, s = self.batch_size, self.spec
        for _ in range(self.steps):
            # (the float math in numpy: on a many-core host a torch CPU elementwise op over a few thousand elements costs milliseconds —
            # `torch.log` of a [512, 13] tensor took 15 ms per batch on the 1-GPU boxes, 10 s of a 12 s candidate fine-tune of the search loop)
            if s.dense_is_zero:
                int_x = torch.zeros(B, s.Fd)
            else:
                int_x = torch.from_numpy(np.log(torch.randint(0, 1000, (B, s.Fd), generator=g).numpy().astype(np.float32) + np.float32(1.0)))
            cat_x = torch.from_numpy(np.stack([torch.randint(0, int(n), (B,), generator=g).numpy() for n in s.tables], axis=1))
            y = torch.from_numpy((torch.rand(B, 1, generator=g).numpy() < 0.25).astype(np.float32))
            yield int_x, cat_x, y


def _synthetic_options(root_dir):
    opts = {"steps": 200, "test_steps": 4, "seed": 1234}
    if ":" in root_dir:
        for kv in root_dir.split(":", 1)[1].split(","):
            if kv:
                k, v = kv.split("=")
                opts[k] = int(v)
    return opts


def _pipes(args, spec: DatasetSpec):
    assert args.validate_split in ["val", "test"], ValueError("Invalid validation split! Should be in ['val', 'test'].")
    from .dist import world_info
    rank, world = world_info()
    if str(args.root_dir).startswith("synthetic"):
        o = _synthetic_options(args.root_dir)
        o["seed"] += 1000 * rank  # every rank of a data-parallel run draws its own share of the global batch
        if o.get("cap"):  # synthetic ids for capped tables (the reference caps by editing MAX_NUM_EMBEDDINGS, config.py:17-19)
            spec = DatasetSpec(spec.name, spec.Fd, spec.Fs, [min(n, o["cap"]) for n in spec.tables], spec.dense_is_zero)
        return ([SyntheticPipe(spec, args.train_batch_size, o["steps"], o["seed"])],
                [SyntheticPipe(spec, args.test_batch_size, o["test_steps"], o["seed"] + 1)], 1, 1)
    shard_dirs = sorted(glob.glob(os.path.join(args.root_dir, "shard-*")))
    all_dirs = shard_dirs
    if world > 1:
        assert len(shard_dirs) >= world, "data-parallel run on %d GPUs needs at least %d shard-* directories" % (world, world)
        shard_dirs = shard_dirs[rank::world]  # the TRAIN shards are split over the ranks; every rank evaluates all test shards
    print("Training directory...", shard_dirs)
    train_file = "train.txt" if args.train_split == "train" else "trainval.txt"
    test_file = "{}.txt".format(args.validate_split)
    shard_cls = TsvShard if os.environ.get("NASREC_TSV_READER", "native") == "python" else NativeTsvShard
    train = [shard_cls(os.path.join(d, train_file), spec, args.train_batch_size) for d in shard_dirs]
    test = [shard_cls(os.path.join(d, test_file), spec, args.test_batch_size) for d in all_dirs]
    return train, test, len(shard_dirs), len(all_dirs)


def get_criteo_kaggle_pipes(args):
    """-> (train pipes, test pipes, number of train workers, number of test workers), data_pipes.py:36-66"""
    return _pipes(args, SPECS["criteo-kaggle"])


def get_avazu_kaggle_pipes(args):
    return _pipes(args, SPECS["avazu"])


def get_kdd_kaggle_pipes(args):
    return _pipes(args, SPECS["kdd"])


GET_PIPES = {"criteo-kaggle": get_criteo_kaggle_pipes, "avazu": get_avazu_kaggle_pipes, "kdd": get_kdd_kaggle_pipes}


def make_loaders(args):
    """train / test loaders of one run (what main_train.py:86-104 builds from the pipes)"""
    train, test, _, _ = GET_PIPES[args.dataset](args)
    threaded = any(isinstance(p, NativeTsvShard) for p in train)
    # batches arrive on the GPU (pinned blocks, second stream) when the run has one: NASREC_STAGE_ON_DEVICE=0 keeps host batches
    dev = getattr(args, "gpu", None) if (torch.cuda.is_available() and os.environ.get("NASREC_STAGE_ON_DEVICE", "1") != "0") else None
    return RoundRobinLoader(train, prefetch=threaded, device=dev), RoundRobinLoader(test, prefetch=threaded, device=dev)
