"""The C multi-GPU entry points (include/vstree_amd_multi.h) on a one-GPU box:
several replicas of the index on device 0, one host thread each.  What is
under test is everything but the wire: the block split with global query
numbers, the concatenation in reference order, the range-partitioned MUM
filter with peer copies and carries -- all through the real GPU kernels
(vsa_result_partition -> exchange -> range filter), compared with the golden
lists of the single-process reference."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def MG(V):
    from vstree_amd import multi
    return multi


def tables(MG, case, devices):
    idx, q = H.load_case(case)
    i = idx.as_width(64)
    return idx, q, MG.Multi.from_tables(i.n, i.prefixlength, i.numofchars,
                                        i.tis, i.suf, i.lcp, i.llv, i.bck,
                                        i.bwt, devices=devices)


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_replicas_reproduce_the_reference_lists(V, MG, world):
    idx, q, m = tables(MG, "c1", [0] * world)
    assert m.ndevices() == world
    for mode, key, L in ((MG.COMPLETE, "complete", 0),
                         (MG.MUMCAND, "mumcand20", 20), (MG.MUM, "mum20", 20),
                         (MG.MEM, "mem20_sp2", 20)):
        got, st, rc, msg = m.findmatches(mode, q.symbols, q.start, q.length, L)
        assert rc == 0, msg
        want = H.expected("c1", key)
        assert np.array_equal(H.matches_as_ref(idx, got), want), (world, key)
        assert st.count == len(want)
        assert st.sumlength == int(want["length"].sum())
        if mode == MG.MUM:
            assert st.candidates == len(H.expected("c1", "mumcand20"))
    # counters of replicas that share a device are summed on the host
    assert m.uses_rccl() == (world == 1)
    m.set_queryspeedup(0)
    got, _, rc, _ = m.findmatches(MG.MEM, q.symbols, q.start, q.length, 20)
    assert np.array_equal(H.matches_as_ref(idx, got),
                          H.expected("c1", "mem20_sp0"))
    m.close()


@pytest.mark.parametrize("world", [1, 3])
def test_replicas_answer_approximate_matching(V, MG, world):
    """vmatch -complete -e K | -h K over replicas: every replica takes a
    block of the reads, the lists in block order are the reference's list
    (golden c1 / c6: pigeonhole path, tree path, percent thresholds); a
    configuration one replica declines is declined for the whole job"""
    for case in ("c1", "c6"):
        idx, q, m = tables(MG, case, [0] * world)
        for key in sorted(H.manifest()[case]["runs"]):
            if not key.startswith("approx_"):
                continue
            spec = key[len("approx_"):]
            doedist, k, pct = spec[0] == "e", int(spec[1:].rstrip("pb")), \
                1 if spec.endswith("p") else (2 if spec.endswith("b") else 0)
            got, st, rc, msg = m.findapproxcompletematches(
                q.symbols, q.start, q.length, doedist, k, pct)
            assert rc == 0, (case, key, msg)
            want = H.expected(case, key)
            assert np.array_equal(H.matches_as_ref(idx, got), want), \
                (world, case, key)
            assert st.count == len(want)
        m.close()
    # the reference's error ends the job at its read; the reads in front of
    # it, also those of earlier blocks, are answered
    idx, q, m = tables(MG, "c5", [0] * world)
    reads = H.Queries.from_list([q.symbols[:150], q.symbols[400:550],
                                 q.symbols[200:203], q.symbols[600:750]])
    got, st, rc, msg = m.findapproxcompletematches(
        reads.symbols, reads.start, reads.length, True, 3)
    assert rc != 0 and "threshold=3>=3=patternlen not allowed" in msg
    first = H.Queries.from_list([q.symbols[:150], q.symbols[400:550]])
    assert np.array_equal(got, H.oracle_approx(idx, first, True, 3))
    m.close()


def test_a_declined_job_gives_its_memory_back_and_replicas_are_alike(V, MG):
    """ a job the engine declines (VSA_NOT_COVERED: reads beyond 512
    symbols in approximate matching) and a job the reference ends with an
    error leave the devices as they found them -- no buffer of an exchange
    that never ran stays behind; and the replicas of a set made from host
    tables are copies of ONE upload: the same derived tables of the same depth
    whatever memory the device had free when each of them arrived"""
    idx, q, m = tables(MG, "c5", [0, 0, 0])
    depths = set()
    for r in range(3):
        h = V.Index(H.C.c_void_p(MG.lib.vsa_multi_index(m._h, r)))
        info = h.info()
        h._h = None                       # borrowed: the set owns it
        depths.add((info.deepprefix, info.device_bytes))
    assert len(depths) == 1
    # warm every path once, then compare free memory around failing jobs
    rng = np.random.default_rng(5)
    longreads = H.Queries.uniform(rng.integers(0, 4, 6 * 600).astype(np.uint8),
                                  600)
    short = H.fasta_queries(H.os.path.join(H.GOLDEN, "short.fna"))
    mixed = H.Queries.from_list([q.symbols[:150], q.symbols[400:550],
                                 q.symbols[200:203], q.symbols[600:750]])

    def failing_jobs():
        got, st, rc, msg = m.findapproxcompletematches(
            longreads.symbols, longreads.start, longreads.length, True, 2)
        assert rc == V.NOT_COVERED, (rc, msg)
        assert len(got) == 0
        got, st, rc, msg = m.findapproxcompletematches(
            mixed.symbols, mixed.start, mixed.length, True, 3)
        assert rc != 0 and "not allowed" in msg
        got, st, rc, msg = m.findmatches(MG.MUM, short.symbols, short.start,
                                         short.length, 3)
        assert rc < 0 and "must be >=" in msg

    failing_jobs()
    before = V.device_meminfo(0)[0]
    for _ in range(3):
        failing_jobs()
    after = V.device_meminfo(0)[0]
    assert after >= before - (1 << 20), (before, after)
    # and the set still answers
    got, st, rc, msg = m.findapproxcompletematches(
        q.symbols, q.start, q.length, True, 4, 2)
    assert rc == 0 and np.array_equal(H.matches_as_ref(idx, got),
                                      H.expected("c5", "approx_e4b"))
    m.close()


def device_blocks(V, q, world):
    """the queries of a case cut into `world` contiguous blocks (the split of
    vsa_multi_findmatches), each uploaded to device 0 with its global offset"""
    blocks, base, extra = [], q.nq // world, q.nq % world
    for r in range(world):
        first = r * base + min(r, extra)
        count = base + (1 if r < extra else 0)
        st, ln = q.start[first:first + count], q.length[first:first + count]
        lo = int(st.min()) if count else 0
        hi = int((st + ln).max()) if count else 0
        b = V.Queries.from_host(q.symbols[lo:hi], st - lo, ln, 0)
        b.set_offset(first)
        blocks.append(b)
    return blocks


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_device_resident_blocks_reproduce_the_reference_lists(V, MG, world):
    """vsa_multi_findmatches_device: the blocks lie in HBM, the lists stay
    there; in replica order they are the reference's list, for -mum each
    replica holds the MUMs of its range of the index"""
    idx, q, m = tables(MG, "c1", [0] * world)
    blocks = device_blocks(V, q, world)
    for mode, key, L in ((MG.COMPLETE, "complete", 0),
                         (MG.MUMCAND, "mumcand20", 20), (MG.MUM, "mum20", 20),
                         (MG.MEM, "mem20_sp2", 20), (MG.MUM, "mum20", 20)):
        res, st, rc, msg = m.findmatches_device(mode, blocks, L)
        assert rc == 0, msg
        assert all(r is not None for r in res)
        lists = [r.fetch() for r in res]
        got = np.concatenate(lists)
        want = H.expected("c1", key)
        assert np.array_equal(H.matches_as_ref(idx, got), want), (world, key)
        assert st.count == len(want) == sum(r.count for r in res)
        assert st.sumlength == int(want["length"].sum())
        if mode == MG.MUM:
            assert st.candidates == len(H.expected("c1", "mumcand20"))
            # replica r holds range r: floor(dbstart * world / (n + 1)) == r
            for r, l in enumerate(lists):
                part = (l["dbstart"].astype(object) * world) // (idx.n + 1)
                assert all(p == r for p in part)
        for r in res:
            r.close()
    assert m.uses_rccl() == (world == 1)
    # a block on the wrong device / a missing block is refused
    res, st, rc, msg = m.findmatches_device(MG.COMPLETE, blocks, 0)
    assert rc == 0
    for r in res:
        r.close()
    m.close()


def test_device_form_errors_follow_the_reference(V, MG):
    idx, _, m = tables(MG, "grumbach", [0, 0, 0])
    q = H.fasta_queries(H.os.path.join(H.GOLDEN, "short.fna"))
    blocks = device_blocks(V, q, 3)
    res, st, rc, msg = m.findmatches_device(MG.COMPLETE, blocks, 0)
    assert rc < 0 and msg == "patternlength=5 must be >= 6=prefixlen"
    got = np.concatenate([r.fetch() for r in res if r is not None])
    assert np.array_equal(H.matches_as_ref(idx, got),
                          H.expected("grumbach", "complete_short"))
    res, st, rc, msg = m.findmatches_device(MG.MUM, blocks, 3)
    assert rc < 0 and msg == "searchlength=3 must be >= 6=prefixlen"
    assert all(r is None for r in res)
    # ragged reads, repeats beyond 255 and wildcards through the exchange
    m.close()
    idx, q, m = tables(MG, "largepat", [0, 0, 0])
    blocks = device_blocks(V, q, 3)
    pl = idx.prefixlength
    for mode, kw in ((MG.MEM, {}), (MG.MUMCAND, dict(mum=True, cand=True)),
                     (MG.MUM, dict(mum=True))):
        res, st, rc, msg = m.findmatches_device(mode, blocks, pl + 6)
        assert rc == 0, msg
        got = np.concatenate([r.fetch() for r in res])
        assert np.array_equal(
            got, H.oracle_querymatches(idx, q, pl + 6, speedup=2, **kw)), mode
    m.close()


@pytest.mark.parametrize("world,per", [(1, 3000), (2, 999), (3, 2500),
                                       (5, 640)])
def test_multi_pipeline_deals_batches_out_and_keeps_the_order(V, MG, world,
                                                              per):
    """vsa_multi_pipeline_*: packed batches from host memory, dealt out to
    the replicas in turn; the lists come back in submission order = query
    order; -mum filters over the candidates of all replicas at the end"""
    idx, q, m = tables(MG, "c1", [0] * world)
    m.set_queryspeedup(2)
    ql = int(q.length[0])
    sym = q.symbols.copy()
    sym[np.random.default_rng(3).integers(0, len(sym), 300)] = H.WILDCARD
    hq = H.Queries.uniform(sym, ql)
    for mode, kw in ((MG.COMPLETE, None), (MG.MUMCAND, dict(mum=True,
                                                            cand=True)),
                     (MG.MEM, dict()), (MG.MUM, dict(mum=True))):
        p = MG.MultiPipeline(m, mode, 20, ql, per, maxspecial=per)
        out, first = [], 0
        while first < hq.nq:
            n = min(per, hq.nq - first)
            while not p.pack_into_slot(sym[first * ql:(first + n) * ql], n):
                rc, got = p.next()
                assert rc == 0
                out.append(got)
            first += n
        while True:
            rc, got = p.next()
            if rc == 1:
                break
            assert rc == 0
            out.append(got)
        got = np.concatenate(out) if out else np.zeros(0, MG.MATCH_DTYPE)
        if mode == MG.MUM:
            assert len(got) == 0
            lists, st = p.finish()
            got = np.concatenate(lists)
            assert st.count == len(got)
            for r, l in enumerate(lists):
                part = (l["dbstart"].astype(object) * world) // (idx.n + 1)
                assert all(x == r for x in part)
        want = (H.oracle_complete(idx, hq) if kw is None else
                H.oracle_querymatches(idx, hq, 20, speedup=2, **kw))
        assert np.array_equal(got, want), (world, per, mode)
        # a second job through the same pipelines: the numbering starts again
        if mode == MG.MUM:
            assert p.pack_into_slot(sym[:per * ql], min(per, hq.nq))
            while p.next()[0] != 1:
                pass
            lists, st = p.finish()
            want = H.oracle_querymatches(
                idx, H.Queries.uniform(sym[:min(per, hq.nq) * ql], ql), 20,
                speedup=2, mum=True)
            assert np.array_equal(np.concatenate(lists), want)
        p.close()
    m.close()


def test_one_replica_sums_its_counters_through_rccl(V, MG):
    """a communicator of one rank: the ncclAllReduce path runs for real"""
    idx, q, m = tables(MG, "micro", [0])
    got, st, rc, msg = m.findmatches(MG.COMPLETE, q.symbols, q.start,
                                     q.length)
    assert rc == 0, msg
    assert m.uses_rccl()
    want = H.oracle_complete(idx, q)
    assert np.array_equal(got, want) and st.count == len(want)
    m.close()


def test_repetitive_text_ragged_queries_and_wildcards(V, MG):
    idx, q, m = tables(MG, "largepat", [0, 0, 0])
    pl = idx.prefixlength
    for mode, kw in ((MG.MEM, {}), (MG.MUMCAND, dict(mum=True, cand=True)),
                     (MG.MUM, dict(mum=True))):
        got, st, rc, msg = m.findmatches(mode, q.symbols, q.start, q.length,
                                         pl + 6)
        assert rc == 0, msg
        assert np.array_equal(
            got, H.oracle_querymatches(idx, q, pl + 6, speedup=2, **kw)), mode
    m.close()
    idx, q, m = tables(MG, "wildcards", [0, 0])
    got, _, rc, _ = m.findmatches(MG.MEM, q.symbols, q.start, q.length, 2)
    assert rc == 0
    assert np.array_equal(H.matches_as_ref(idx, got),
                          H.expected("wildcards", "mem2"))
    m.close()


def test_replicate_copies_an_index_device_to_device(V, MG):
    idx, q = H.load_case("grumbach")
    i = idx.as_width(32)
    first = V.Index.from_tables(i.n, i.prefixlength, i.numofchars, i.tis,
                                i.suf, i.lcp, i.llv, i.bck, i.bwt)
    single = V.findquerymatches(first, V.Queries.from_host(
        q.symbols, q.start, q.length), 14, mum=True).fetch()
    # a clone answers like the original
    twin = first.clone(0)
    assert twin.info().device_bytes == first.info().device_bytes
    assert np.array_equal(V.findquerymatches(twin, V.Queries.from_host(
        q.symbols, q.start, q.length), 14, mum=True).fetch(), single)
    twin.close()
    m = MG.Multi.replicate(first, [0, 0, 0, 0])
    got, st, rc, msg = m.findmatches(MG.MUM, q.symbols, q.start, q.length, 14)
    assert rc == 0, msg
    assert np.array_equal(got, single)
    assert np.array_equal(H.matches_as_ref(idx, got),
                          H.expected("grumbach", "mum14"))
    m.close()


def test_errors_and_callbacks_follow_the_reference(V, MG):
    idx, _, m = tables(MG, "grumbach", [0, 0, 0])
    q = H.fasta_queries(H.os.path.join(H.GOLDEN, "short.fna"))
    got, st, rc, msg = m.findmatches(MG.COMPLETE, q.symbols, q.start,
                                     q.length)
    # the short query stops the run after the matches before it
    assert rc < 0 and msg == "patternlength=5 must be >= 6=prefixlen"
    assert np.array_equal(H.matches_as_ref(idx, got),
                          H.expected("grumbach", "complete_short"))
    got, st, rc, msg = m.findmatches(MG.MEM, q.symbols, q.start, q.length, 3)
    assert rc < 0 and msg == "searchlength=3 must be >= 6=prefixlen"
    _, q2 = H.load_case("grumbach")
    want = H.oracle_querymatches(idx, q2, 14, speedup=2)
    rc, got = m.findmatches_cb(MG.MEM, q2.symbols, q2.start, q2.length, 14)
    assert rc == 0 and got == [tuple(int(x) for x in r) for r in want.tolist()]
    rc, got = m.findmatches_cb(MG.MEM, q2.symbols, q2.start, q2.length, 14,
                               stop_after=7)
    assert rc == -1 and len(got) == 7
    # no queries at all
    e = np.zeros(0, np.uint64)
    got, st, rc, _ = m.findmatches(MG.MUM, np.zeros(0, np.uint8), e, e, 14)
    assert rc == 0 and len(got) == 0 and st.count == 0
    m.close()


def test_two_processes_on_one_gpu_run_the_exchange_for_real(tmp_path):
    """bench.py's N > 1 path with TWO processes on the one GPU of the box:
    vsa_findmumcandidates_packed -> vsa_result_partition -> exchange (gloo on
    host copies, the box has no second GPU for RCCL) ->
    vsa_mumuniqueinquery_range_packed with carries; the job-wide counters must
    be those of one process answering all queries."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    common = ["--genome", "3e7", "--steps", "2", "--warmup", "1", "--quick",
              "--cpu-sample", "0"]
    bench = H.os.path.join(H.ROOT, "bench.py")
    two = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), bench, "--gpus", "2",
         "--rehearse-on-one-gpu", "--queries", "150000"] + common,
        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert two.returncode == 0, two.stderr.decode()[-2000:]
    one = subprocess.run([sys.executable, bench, "--queries", "300000"] +
                         common, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    a = json.loads(two.stdout.decode().strip().splitlines()[-1])
    b = json.loads(one.stdout.decode().strip().splitlines()[-1])
    assert a["n_gpus"] == 2 and b["n_gpus"] == 1
    assert a["matches"] == b["matches"] > 250000
    assert a["candidates"] == b["candidates"] >= a["matches"]
    assert a["query_suffix_searches"] == b["query_suffix_searches"]


def test_bench_started_directly_with_two_gpus_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how a driver
    starts the scaling run) must end up on two ranks, or not print a line at
    all -- never a line for one GPU."""
    import json
    import subprocess
    import sys
    bench = H.os.path.join(H.ROOT, "bench.py")
    common = ["--genome", "3e7", "--queries", "150000", "--steps", "2",
              "--warmup", "1", "--quick", "--cpu-sample", "0"]
    two = subprocess.run([sys.executable, bench, "--gpus", "2",
                          "--rehearse-on-one-gpu"] + common,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=600)
    assert two.returncode == 0, two.stderr.decode()[-2000:]
    lines = [l for l in two.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    a = json.loads(lines[0])
    assert a["n_gpus"] == 2 and a["ranks"] == 2
    assert a["rccl_ranks"] == 0      # gloo on host copies in the rehearsal
    # more GPUs than the box has, no rehearsal: refusal, no line
    many = subprocess.run([sys.executable, bench, "--gpus", "7"] + common,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=600)
    assert many.returncode != 0 and many.stdout.decode().strip() == ""
    # a rank count that does not match --gpus: refusal as well
    env = dict(H.os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    odd = subprocess.run([sys.executable, bench, "--gpus", "2"] + common,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=600, env=env)
    assert odd.returncode != 0 and odd.stdout.decode().strip() == ""


def test_bench_c_path_with_replicas_on_one_gpu():
    """`bench.py --gpus 2` takes the product's N > 1 entry by default (--path
    c: vsa_multi_findmatches_device from one process), here with two replicas
    on device 0; `--host` times vsa_multi_findmatches instead; counters = one
    GPU's on the same queries."""
    import json
    import subprocess
    import sys
    bench = H.os.path.join(H.ROOT, "bench.py")
    common = ["--genome", "3e7", "--steps", "2", "--warmup", "1", "--quick",
              "--cpu-sample", "0"]

    def run(*args):
        c = subprocess.run([sys.executable, bench] + list(args) + common,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=600)
        assert c.returncode == 0, c.stderr.decode()[-2000:]
        lines = [l for l in c.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1
        return json.loads(lines[0])

    a = run("--gpus", "2", "--replicas-on-one-gpu", "--queries", "150000")
    h = run("--gpus", "2", "--path", "c", "--host", "--replicas-on-one-gpu",
            "--queries", "150000")
    b = run("--queries", "300000")
    assert a["n_gpus"] == 2 and "resident in HBM" in a["config"]["path"]
    assert "30 Mbp" in a["config"]["workload"]
    assert h["n_gpus"] == 2 and "host memory" in h["metric"]
    assert "vsa_multi_pipeline" in h["config"]["path"]
    c = run("--gpus", "2", "--path", "c", "--host", "--compat",
            "--replicas-on-one-gpu", "--queries", "150000")
    assert "vsa_multi_findmatches" in c["config"]["path"]
    for x in (a, h, c):
        assert x["matches"] == b["matches"] > 250000
        assert x["candidates"] == b["candidates"]
    for x in (a, c):
        assert x["query_suffix_searches"] == b["query_suffix_searches"]


def test_bench_c_path_under_the_drivers_launcher():
    """how the driver starts the scaling run: `python -m torch.distributed.run
    --nproc-per-node N bench.py --gpus N`.  The C path runs in rank 0 (one
    process, a host thread per GPU, no torch in it: one HIP runtime per
    process), the other ranks leave; ONE line for N GPUs comes out.  And where
    the C path fails the job is not lost: rank 0 starts the other N > 1 form
    and the line says so (rehearsed here with two replicas / two gloo ranks on
    the one GPU)."""
    import json
    import socket
    import subprocess
    import sys
    bench = H.os.path.join(H.ROOT, "bench.py")
    common = ["--gpus", "2", "--replicas-on-one-gpu", "--genome", "3e7",
              "--queries", "150000", "--steps", "2", "--warmup", "1",
              "--quick", "--cpu-sample", "0"]

    def launch(*more):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        c = subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
             "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(port), bench] + common + list(more),
            stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert c.returncode == 0, c.stderr.decode()[-3000:]
        lines = [l for l in c.stdout.decode().splitlines()
                 if l.startswith("{")]
        assert len(lines) == 1, c.stdout.decode()[-2000:]
        return json.loads(lines[0]), c.stderr.decode()

    a, err = launch()
    assert a["n_gpus"] == 2 and a["launcher_ranks"] == 2
    assert "resident in HBM" in a["config"]["path"]
    assert "c_path_failed" not in a
    assert "rank 1 of 2: the C path runs in rank 0" in err
    b, err = launch("--test-fail-c-path")
    assert b["n_gpus"] == 2 and b["ranks"] == 2
    assert b["c_path_failed"] == "RuntimeError: --test-fail-c-path"
    assert b["matches"] == a["matches"] > 250000
    assert b["candidates"] == a["candidates"]


def test_queries_in_any_order_and_outside_the_buffer(V, MG):
    """a Multiseq's queries need not lie in ascending order in the buffer
    (vsa_queries_from_host takes any order): every block is uploaded from the
    lowest start to the highest end of ITS queries; a query that reaches
    beyond nsymbols is refused before anything is read"""
    idx, q, m = tables(MG, "c1", [0, 0, 0])
    rng = np.random.default_rng(5)
    perm = rng.permutation(q.nq)
    # the same symbols, the queries listed in another order: query i of the
    # call is query perm[i] of the golden case
    start, length = q.start[perm], q.length[perm]
    got, st, rc, msg = m.findmatches(MG.COMPLETE, q.symbols, start, length)
    assert rc == 0, msg
    want = H.oracle_complete(idx, H.Queries(q.symbols, start, length))
    assert np.array_equal(got.view(np.uint64).reshape(-1, 4),
                          np.asarray(want).view(np.uint64).reshape(-1, 4))
    got, st, rc, msg = m.findmatches(MG.MUM, q.symbols, start, length, 20)
    assert rc == 0, msg
    want = H.oracle_querymatches(idx, H.Queries(q.symbols, start, length), 20,
                                 mum=True)
    assert np.array_equal(got.view(np.uint64).reshape(-1, 4),
                          np.asarray(want).view(np.uint64).reshape(-1, 4))
    bad = length.copy()
    bad[7] = len(q.symbols)          # runs past the end of the buffer
    got, st, rc, msg = m.findmatches(MG.COMPLETE, q.symbols, start, bad)
    assert rc == -2 and "query 7" in msg and len(got) == 0
    m.close()
