"""`gloo` tests of the N>1 path on CPU at world sizes 2, 3 and 8 (8 = the node BASELINE.json names; 3 = a size that
divides nothing): contiguous candidate shards need no collective for inference; training = one all-reduce(SUM) of
the flat gradient per step.  Sizes below the world size give ranks with EMPTY shards."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_everything():
    from clairvoyante_amd import parallel
    for total in (0, 1, 7, 8, 9, 40000000, 4194304 + 3):
        for ws in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_a_mistyped_exchange_plan_is_refused_everywhere(monkeypatch):
    """CV_EXCHANGE is read by plan_exchange AND, for a model nobody planned, by exchange_mode: a typo must not fall
    through to the two-piece exchange silently on either path"""
    from clairvoyante_amd import parallel

    class Stub(object):
        device = torch.device("cpu")
    monkeypatch.setenv("CV_EXCHANGE", "onee")
    with pytest.raises(ValueError):
        parallel.exchange_mode(Stub())
    with pytest.raises(ValueError):
        parallel.plan_exchange(Stub(), 10000)
    monkeypatch.setenv("CV_EXCHANGE", "one")
    assert parallel.exchange_mode(Stub()) == "one"
    monkeypatch.delenv("CV_EXCHANGE")
    assert parallel.exchange_mode(Stub()) == "split"


def _worker(rank, ws, port, tmp, n):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    from clairvoyante_amd import parallel, synth
    from oracle import cv_oracle as O
    import common
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, ws)
    arch = "slim"
    P = common.bench_params(O, arch)
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=4, return_class=True)
    y = synth.make_labels(cls, rf, alt, il).numpy(); x = xt.numpy()
    lo, hi = parallel.shard_range(n, rank, ws)
    # inference shard: rank-local, concatenation in rank order == the unsharded result
    out = O.predict(arch, P, x[lo:hi])
    gathered = [None] * ws
    dist.all_gather_object(gathered, out)
    full = O.predict(arch, P, x)
    assert np.array_equal(np.concatenate(gathered), full)
    # training: shard gradients (data terms), all-reduce SUM, add lambda*w once
    lam = 0.01
    l_sh, parts, g_sh = O.loss_grad(arch, P, x[lo:hi], y[lo:hi], lam=0.0)
    flat = torch.from_numpy(np.concatenate([g_sh[k].ravel() for k in O.PARAM_NAMES]))
    l_all, parts_all, g_all = O.loss_grad(arch, P, x, y, lam=lam)
    ref = np.concatenate([(g_all[k] - (lam * P[k] if "bias" not in k else 0)).ravel() for k in O.PARAM_NAMES])
    # the product's exchange (parallel.exchange_bucket) on a bucket laid out as cv_grad_async leaves it
    # (include/clairvoyante_amd.h: 16-float loss header of (hi, lo) pairs + flat gradient), two collectives
    l2 = lam * sum(float(np.sum(P[k].astype(np.float64) ** 2)) / 2 for k in O.PARAM_NAMES if "bias" not in k)
    hdr = np.zeros(16, dtype=np.float32)
    for k, d in enumerate(list(parts[0:4]) + [l2]):
        hdr[2 * k] = np.float32(d); hdr[2 * k + 1] = np.float32(d - float(hdr[2 * k]))
    hdr[10] = 1.0

    class Stub(object):
        device = torch.device("cpu")
    m = Stub()
    m._bucket = torch.cat([torch.from_numpy(hdr), flat])
    m._bucket_header = 16
    m._bucket_dense = 16 + sum(P[k].size for k in O.PARAM_NAMES[:6])
    assert parallel.comm_stream(m) is None
    if ws != 2:       # the plan for a tiny share: ONE collective over the whole bucket (ws == 2: the unplanned default, two pieces)
        assert parallel.plan_exchange(m, n) == "one" and parallel.exchange_mode(m) == "one"
    else:
        assert parallel.exchange_mode(m) == "split"
    parallel.exchange_bucket(m, None)
    got = m._bucket.numpy()
    assert np.abs(got[16:] - ref).max() <= 1e-4 * np.abs(ref).max()      # shard gradients sum to the whole-batch gradient
    dec = [float(got[2 * k]) + float(got[2 * k + 1]) for k in range(5)]
    assert np.allclose(dec[0:4], parts_all[0:4], rtol=1e-7)
    assert got[10] == ws and abs(dec[4] / got[10] - l2) <= 5e-7 * l2      # fp32 sum of ws equal (hi, lo) pairs: exact at powers of two only
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")


@pytest.mark.parametrize("ws,n", [(2, 22), (3, 22), (8, 22), (8, 5), (3, 1)])
def test_gloo_shards_and_gradient_allreduce(tmp_path, oracle, ws, n):
    """(8, 5) and (3, 1): more ranks than candidates -- the ranks with an empty shard contribute a zero gradient and
    zero losses and still take part in both collectives"""
    port = 29600 + (os.getpid() + 31 * ws + n) % 300
    mp.spawn(_worker, args=(ws, port, str(tmp_path), n), nprocs=ws, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(ws))


# ---- callVar under two ranks: block-cyclic split of the input lines, per-rank record fragments, merge -------------

def _write_tensor_text(path, x, seed=3, bad_every=97):
    import gzip
    rng = np.random.RandomState(seed)
    raw = x.copy()
    for i in range(1, 4):
        raw[:, :, :, i] += raw[:, :, :, 0]
    with gzip.open(path, "wt") as f:
        for j in range(raw.shape[0]):
            seq = "".join(rng.choice(list("ACGT"), 33))
            if j % bad_every == 5:
                seq = seq[:16] + "N" + seq[17:]          # dropped by GetTensor (utils_v2.py:38-40)
            f.write("%s %d %s %s\n" % ("chr%d" % (1 + j % 4), 10000 + 7 * j, seq,
                                       " ".join("%0.1f" % v for v in raw[j].reshape(-1))))


def _callvar_worker(rank, ws, port, tmp, tfn, block_lines):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    import io
    import types
    from clairvoyante_amd import callVar, parallel, utils_v2
    from oracle import cv_oracle as O
    import common
    parallel.init_from_env(backend="gloo")
    P = common.bench_params(O, "slim")
    args = types.SimpleNamespace(v2=False, v3=True, showRef=False, qual=30, ref_fn=None, sampleName="S")
    call_fn = os.path.join(tmp, "calls.vcf")
    frag = open("%s.rank%d" % (call_fn, rank), "w")
    index = []
    files = tfn.split(",")
    src = utils_v2.GetTensorFiles(files, 37, rank, ws) if len(files) > 1 else utils_v2.GetTensorBlocks(tfn, block_lines, rank, ws)
    for block, c, X, pos in src:
        assert block % ws == rank
        buf = io.StringIO()
        if c:
            o = O.predict("slim", P, X)
            callVar.Output(args, buf, c, X, pos, o[:, 0:4], o[:, 4:6], o[:, 6:10], o[:, 10:16])
        frag.write(buf.getvalue())
        nb = len(buf.getvalue().encode("ascii"))
        if index and index[-1][0] == block:        # a file comes in several batches: one index entry (callVar.TestSharded)
            index[-1] = (block, index[-1][1] + nb)
        else:
            index.append((block, nb))
    frag.close()
    with open("%s.rank%d.idx" % (call_fn, rank), "w") as f:
        f.write("".join("%d %d\n" % e for e in index))
    dist.barrier()
    if rank == 0:
        with open(call_fn, "w") as fh:
            callVar.PrintVCFHeader(args, fh)
            callVar.merge_fragments(call_fn, ws, fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ws,n,block_lines", [(2, 700, 64), (2, 130, 64), (2, 64, 64), (2, 5, 64), (2, 300, 1000),
                                              (3, 700, 64), (8, 700, 64), (8, 130, 64), (8, 5, 64), (3, 64, 64)])
def test_callvar_split_and_merge_equals_one_process(oracle, tmp_path, ws, n, block_lines):
    """the sharding / merge logic of callVar.TestSharded with the oracle in the model's place (no GPU here): every
    rank parses only its blocks of input lines, formats its records, rank 0 joins the fragments in input order --
    the file must equal the single-process VCF byte for byte, whatever the number of blocks (fewer blocks than
    ranks, a ragged last block, rows dropped for a non-ACGT centre base)"""
    import io
    import types
    import common
    from clairvoyante_amd import callVar, utils_v2
    x = common.inputs(n, seed=23)
    tfn = str(tmp_path / "t.gz")
    _write_tensor_text(tfn, x)
    port = 29700 + (os.getpid() + n + 17 * ws) % 200
    mp.spawn(_callvar_worker, args=(ws, port, str(tmp_path), tfn, block_lines), nprocs=ws, join=True)
    got = open(str(tmp_path / "calls.vcf")).read()
    args = types.SimpleNamespace(v2=False, v3=True, showRef=False, qual=30, ref_fn=None, sampleName="S")
    P = common.bench_params(oracle, "slim")
    fh = io.StringIO()
    callVar.PrintVCFHeader(args, fh)
    for end, c, X, pos in utils_v2.GetTensor(tfn, 100, log=False):
        if c:
            o = oracle.predict("slim", P, X)
            callVar.Output(args, fh, c, X, pos, o[:, 0:4], o[:, 4:6], o[:, 6:10], o[:, 10:16])
    assert got == fh.getvalue()
    assert got.count("\n") > 10 or n < 20
    assert not [f for f in os.listdir(str(tmp_path)) if ".rank" in f]          # fragments are removed


@pytest.mark.parametrize("ws,sizes", [(2, (120, 0, 75)), (2, (40,)), (2, (30, 50, 20, 10, 60)),
                                      (3, (30, 50, 20, 10, 60)), (8, (40, 12, 0, 33, 25, 18, 7, 21, 9, 30, 14)), (8, (40, 9))])
def test_callvar_over_a_list_of_files_equals_one_process_per_file(oracle, tmp_path, ws, sizes):
    """--tensor_fn a.gz,b.gz,...: file k belongs to rank k % ws (no rank inflates another rank's file); the merged VCF is
    the concatenation, in list order, of what one process writes for each file (an empty file, fewer files than ranks)"""
    import io
    import types
    import common
    from clairvoyante_amd import callVar, utils_v2
    files = []
    for k, n in enumerate(sizes):
        fn = str(tmp_path / ("t%d.gz" % k))
        _write_tensor_text(fn, common.inputs(max(n, 1), seed=31 + k)[:n], seed=5 + k)
        files.append(fn)
    port = 29750 + (os.getpid() + len(sizes) * 7 + 17 * ws) % 200
    mp.spawn(_callvar_worker, args=(ws, port, str(tmp_path), ",".join(files), 64), nprocs=ws, join=True)
    got = open(str(tmp_path / "calls.vcf")).read()
    args = types.SimpleNamespace(v2=False, v3=True, showRef=False, qual=30, ref_fn=None, sampleName="S")
    P = common.bench_params(oracle, "slim")
    fh = io.StringIO()
    callVar.PrintVCFHeader(args, fh)
    for fn in files:
        for end, c, X, pos in utils_v2.GetTensor(fn, 100, log=False):
            if c:
                o = oracle.predict("slim", P, X)
                callVar.Output(args, fh, c, X, pos, o[:, 0:4], o[:, 4:6], o[:, 6:10], o[:, 10:16])
    assert got == fh.getvalue()
