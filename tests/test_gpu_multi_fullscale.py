"""BASELINE.json configs[3] at full per-rank size on the one GPU of the box:
a 3 Gbp index, TWO replicas of it on device 0 (deep prefix 15: 2 x 63.5 GB)
and 2 x 10 M reads of 100 bp through the product's N > 1 entry,
vsa_multi_findmatches_device -- blocks resident in HBM, lists left there,
-mum candidates exchanged as 16-byte rows between the replicas and filtered
per range with carries.  -complete, -mum cand and -mum must give the lists of
ONE replica answering all 20 M reads, and the job's counters must be the ones
a one-replica set sums through its RCCL communicator.
Loops the path shards: Vmengine/fquery.c:468-486, kurtz/cleanMUMcand.c:55-118.
VSA_FULLSCALE_BP / VSA_FULLSCALE_QUERIES run it at another size."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

N = int(float(os.environ.get("VSA_FULLSCALE_BP", "3e9")))
NQ = int(float(os.environ.get("VSA_FULLSCALE_QUERIES", "1e7")))
M, L = 100, 20
WORLD = 2


@pytest.fixture(scope="module")
def job(V):
    from vstree_amd import multi as MG
    old = os.environ.get("VSA_DEEP_PREFIX")
    os.environ["VSA_DEEP_PREFIX"] = "15"
    try:
        dg = V.device_malloc(N + 64)
        V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, N, dg, 0))
        index = V.Index.build_device(dg, N, 4, 0)
    finally:
        if old is None:
            del os.environ["VSA_DEEP_PREFIX"]
        else:
            os.environ["VSA_DEEP_PREFIX"] = old
    pos, sub, step = V.synth_query_plan(N, NQ * WORLD, M)
    dq = V.device_malloc(NQ * WORLD * M + 64)
    V._check(V.lib.vsa_synth_queries_device(dg, N, pos.ctypes.data,
                                            sub.ctypes.data, step.ctypes.data,
                                            NQ * WORLD, M, dq, 0))
    V.device_free(dg)
    whole = V.Queries.from_device(dq, NQ * WORLD, M)
    blocks = []
    for r in range(WORLD):
        b = V.Queries.from_device(
            H.C.c_void_p(dq.value + r * NQ * M), NQ, M)
        b.set_offset(r * NQ)
        blocks.append(b)
    V.device_free(dq)
    return dict(MG=MG, index=index, whole=whole, blocks=blocks,
                info=index.info())


def test_two_replicas_of_3_gbp_answer_like_one(V, job):
    MG, index = job["MG"], job["index"]
    assert job["info"].deepprefix == 15 or N < 2e9
    modes = ((MG.COMPLETE, "complete", 0), (MG.MUMCAND, "mum cand", L),
             (MG.MUM, "mum", L))
    # one replica, all 20 M reads: the lists to reproduce ...
    want = {}
    for mode, name, sl in modes:
        if mode == MG.COMPLETE:
            r = V.findcompletematches(index, job["whole"])
        else:
            r = V.findquerymatches(index, job["whole"], sl, mum=True,
                                   cand=(mode == MG.MUMCAND))
        want[mode] = (r.fetch(), r.stats())
        r.close()
    assert len(want[MG.COMPLETE][0]) > 0.7 * NQ * WORLD
    assert len(want[MG.MUM][0]) > NQ * WORLD
    # ... and the counters as a set of ONE replica sums them through RCCL
    one = MG.Multi.replicate(index.clone(0), [0])
    rccl = {}
    for mode, name, sl in modes:
        res, st, rc, msg = one.findmatches_device(mode, [job["whole"]], sl)
        assert rc == 0, msg
        assert one.uses_rccl()
        assert np.array_equal(res[0].fetch(), want[mode][0]), name
        rccl[mode] = (st.count, st.sumlength, st.searches, st.candidates)
        res[0].close()
    one.close()
    V.lib.vsa_device_trim(0)
    two = MG.Multi.replicate(index, [0] * WORLD)
    for mode, name, sl in modes:
        res, st, rc, msg = two.findmatches_device(mode, job["blocks"], sl)
        assert rc == 0, msg
        lists = [r.fetch() for r in res]
        for r in res:
            r.close()
        got = np.concatenate(lists)
        assert len(got) == len(want[mode][0]), name
        assert np.array_equal(got, want[mode][0]), name
        assert (st.count, st.sumlength, st.searches, st.candidates) == \
            rccl[mode], name
        assert st.count == want[mode][1].count
        assert st.sumlength == want[mode][1].sumlength
        if mode == MG.MUM:
            # replica r holds its range of the index
            for r, l in enumerate(lists):
                assert len(l) > 0
                lo, hi = int(l["dbstart"][0]), int(l["dbstart"][-1])
                assert lo * WORLD // (N + 1) == r == hi * WORLD // (N + 1)
    two.close()
