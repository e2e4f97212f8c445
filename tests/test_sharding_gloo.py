"""The N > 1 path on CPU: two ranks over gloo run the rank logic of
vstree_amd/sharding.py.  The per-rank search is done by the CPU oracle here
(the GPU kernels are covered by the -m gpu tests); what is under test is the
sharding, the global query numbering, the candidate exchange and the counter
reduction -- they must reproduce the reference's single-process output."""
import os
import sys

import numpy as np
import pytest

import helpers as H

WORLD = 2


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (H.ROOT, os.path.join(H.ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from vstree_amd import sharding as S
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx, q = H.load_case("c1")
    first, count = S.shard_range(q.nq, rank, world)
    mine = H.Queries(q.symbols, q.start[first:first + count],
                     q.length[first:first + count])
    # phase 1 (no communication): candidates of this rank's queries, numbered
    # globally like vsa_queries_set_offset does
    cand = H.oracle_querymatches(idx, mine, 20, mum=True, cand=True,
                                 speedup=0)
    cand["queryseq"] += np.uint64(first)
    compl = H.oracle_complete(idx, mine)
    compl["queryseq"] += np.uint64(first)
    result = {}

    def filter_fn(allc):
        arr = S.tensor_to_matches(allc, H.MATCH_DTYPE)
        mums = H.oracle_mumfilter(arr)
        result["mums"] = mums
        return len(mums), int(mums["length"].sum())

    local = S.matches_to_tensor(torch, cand)
    nmum, sumlen, ncand = S.global_mum_filter(dist, torch, local, "cpu",
                                              filter_fn)

    # the scalable variant: candidates range-partitioned by dbstart, every
    # rank filters its range with the carry of the lower ranges
    def range_filter_fn(part, carry):
        arr = S.tensor_to_matches(part, H.MATCH_DTYPE)
        mums = H.oracle_mumfilter(arr, carry)
        result["mymums"] = mums
        return len(mums), int(mums["length"].sum())

    pn, ps, pc = S.partitioned_mum_filter(dist, torch, local, idx.n, "cpu",
                                          range_filter_fn)
    assert (pn, ps, pc) == (nmum, sumlen, ncand)
    # the form bench.py uses: candidates in ANY order, grouped by receiving
    # rank beforehand (what vsa_result_partition does on the GPU)
    shuffled = cand[np.random.default_rng(rank).permutation(len(cand))]
    dest = (shuffled["dbstart"] * np.uint64(world)) // np.uint64(idx.n + 1)
    grouped = shuffled[np.argsort(dest, kind="stable")]
    send = np.bincount(dest.astype(np.int64), minlength=world)
    right = (shuffled["dbstart"] + shuffled["length"] - 1).astype(np.int64)
    top = [int(right[dest == r].max()) if (dest == r).any() else 0
           for r in range(world)]
    qn, qs_, qc = S.partitioned_mum_filter_presorted(
        dist, torch, S.matches_to_tensor(torch, grouped), send, top, "cpu",
        range_filter_fn)
    assert (qn, qs_, qc) == (nmum, sumlen, ncand)
    # ... and with the candidates as (key, value) pairs of 16 bytes, the rows
    # vsa_findmumcandidates_packed + vsa_result_partition produce
    bits = 7
    rows = S.pack_candidates(grouped, bits)
    assert np.array_equal(S.unpack_candidates(rows, bits, H.MATCH_DTYPE),
                          grouped)

    def packed_filter_fn(part, carry):
        arr = S.unpack_candidates(part.numpy().astype(np.uint64), bits,
                                  H.MATCH_DTYPE)
        mums = H.oracle_mumfilter(arr, carry)
        assert np.array_equal(mums, result["mymums"])
        return len(mums), int(mums["length"].sum())

    rn, rs, rc_ = S.partitioned_mum_filter_presorted(
        dist, torch, torch.from_numpy(rows.astype(np.int64).reshape(-1)),
        send, top, "cpu", packed_filter_fn, words=2)
    assert (rn, rs, rc_) == (nmum, sumlen, ncand)
    # ... and with the rows for the rank itself behind all others
    # (vsa_result_partition_own): they do not enter the exchange, the filter
    # gets them and the received rows as two lists
    gdest = (grouped["dbstart"] * np.uint64(world)) // np.uint64(idx.n + 1)
    ownlast = np.concatenate([rows[gdest != rank], rows[gdest == rank]])

    def two_lists_filter_fn(own, received, carry):
        assert own.numel() // 2 == int(send[rank])
        return packed_filter_fn(torch.cat([received, own]), carry)

    on, os_, oc = S.partitioned_mum_filter_presorted(
        dist, torch, torch.from_numpy(ownlast.astype(np.int64).reshape(-1)),
        send, top, "cpu", two_lists_filter_fn, words=2, own_last=True)
    assert (on, os_, oc) == (nmum, sumlen, ncand)
    pparts, _ = S.all_gather_matches(
        dist, torch, S.matches_to_tensor(torch, result["mymums"]), "cpu")
    totals = S.all_reduce_counters(dist, torch,
                                   [len(compl), int(compl["length"].sum())],
                                   "cpu")
    # -complete / candidates: concatenation in rank order = reference order
    parts, counts = S.all_gather_matches(dist, torch,
                                         S.matches_to_tensor(torch, compl),
                                         "cpu")
    if rank == 0:
        allcompl = S.tensor_to_matches(torch.cat(parts), H.MATCH_DTYPE)
        np.savez(os.path.join(outdir, "out.npz"), mums=result["mums"],
                 pmums=S.tensor_to_matches(torch.cat(pparts), H.MATCH_DTYPE),
                 compl=allcompl, nmum=nmum, sumlen=sumlen, ncand=ncand,
                 totals=np.array(totals))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_single_process_reference(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    H.load_case("c1")          # warm the oracle build before forking
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD,
             join=True)
    out = np.load(os.path.join(str(tmp_path), "out.npz"))
    idx, q = H.load_case("c1")
    want_mum = H.expected("c1", "mum20")
    got_mum = H.matches_as_ref(idx, out["mums"])
    assert np.array_equal(got_mum, want_mum)
    # range-partitioned filter: rank order = dbstart order, same list
    assert np.array_equal(H.matches_as_ref(idx, out["pmums"]), want_mum)
    assert int(out["nmum"]) == len(want_mum) == 10323
    assert int(out["sumlen"]) == int(want_mum["length"].sum())
    assert int(out["ncand"]) == len(H.expected("c1", "mumcand20"))
    want_c = H.expected("c1", "complete")
    assert np.array_equal(H.matches_as_ref(idx, out["compl"]), want_c)
    assert out["totals"].tolist() == [len(want_c),
                                      int(want_c["length"].sum())]


def _selfmum_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (H.ROOT, os.path.join(H.ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from vstree_amd import sharding as S
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx, _ = H.load_case("grumbach_all")
    first, last = S.selfmum_range(idx.n, rank, world)
    mine = H.selfmum_scan_range(idx, 14, first, last)
    totals = S.all_reduce_counters(dist, torch,
                                   [len(mine), int(mine["length"].sum())],
                                   "cpu")
    parts, _ = S.all_gather_matches(dist, torch,
                                    S.matches_to_tensor(torch, mine), "cpu")
    if rank == 0:
        np.savez(os.path.join(outdir, "selfmum.npz"),
                 mums=S.tensor_to_matches(torch.cat(parts), H.MATCH_DTYPE),
                 totals=np.array(totals))
    dist.barrier()
    dist.destroy_process_group()


def test_self_index_scan_split_over_two_ranks(tmp_path):
    """SURVEY 8e, third row: the scan of fmumself.c split into suffix-array
    ranges, one per rank; rank order = suffix-array order = reference order;
    only the counters are reduced."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    idx, _ = H.load_case("grumbach_all")
    mp.spawn(_selfmum_worker, args=(WORLD, port, str(tmp_path)),
             nprocs=WORLD, join=True)
    out = np.load(os.path.join(str(tmp_path), "selfmum.npz"))
    want = H.expected("grumbach_all", "selfmum14")
    assert np.array_equal(H.selfmatches_as_ref(idx, out["mums"]), want)
    assert out["totals"].tolist() == [len(want), int(want["length"].sum())]
    # the range restatement is the oracle's scan when it covers everything
    assert np.array_equal(H.selfmum_scan_range(idx, 14),
                          H.oracle_selfmum(idx, 14))


def test_selfmum_ranges_tile_the_scan():
    from vstree_amd import sharding as S
    for n in (2, 3, 4, 10, 1000003):
        for world in (1, 2, 3, 8):
            r = [S.selfmum_range(n, k, world) for k in range(world)]
            assert r[0][0] == 2 and r[-1][1] == max(n, 2)
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))


def test_shard_range_covers_everything():
    from vstree_amd import sharding as S
    for total in (0, 1, 7, 10, 1000003):
        for world in (1, 2, 3, 8):
            blocks = [S.shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0
            for (f, c), (f2, _) in zip(blocks, blocks[1:]):
                assert f + c == f2
            assert blocks[-1][0] + blocks[-1][1] == total
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
