"""Rank logic of the multi-GPU query path (one process per GPU, index
replicated, queries sharded), written against torch.distributed so that the
same code runs over RCCL on MI355X (backend "nccl") and over gloo on CPU in
the tests.

Which step needs which collective:
  -complete, -l, -mum cand   none on the data path; one all_reduce(sum) of
                             the match counters
  -mum (not cand)            the candidates of all ranks pass ONE global
                             uniqueness filter (kurtz/cleanMUMcand.c:55-118 of
                             the reference): all_gather of the candidate
                             lists, filter on rank 0
"""
import os

import numpy as np

MATCH_WORDS = 4   # vsa_match = 4 x uint64


def is_rccl(dist):
    """the process group runs on RCCL (torch names it "nccl"; a group made
    with a device map reports e.g. "cpu:gloo,cuda:nccl")"""
    return "nccl" in str(dist.get_backend())


def shard_range(total, rank, world):
    """contiguous block of rank: (first, count); blocks differ by at most 1"""
    base, extra = divmod(int(total), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def selfmum_range(totallength, rank, world):
    """Part of the self-index MUM scan (Vmengine/fmumself.c:33: i = 2 ..
    totallength-1) rank scans: (first, last), last exclusive.  The ranges tile
    the loop; each rank reads lcp/bwt entries i-2 .. i around its ends from its
    own replica of the index (the halo of SURVEY 8e)."""
    first, count = shard_range(max(int(totallength) - 2, 0), rank, world)
    return 2 + first, 2 + first + count


def all_reduce_counters(dist, torch, values, device):
    """sum of a handful of uint64 counters over all ranks"""
    t = torch.tensor([int(v) for v in values], dtype=torch.int64,
                     device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def all_gather_matches(dist, torch, local, device, words=MATCH_WORDS):
    """local: int64 tensor [count*words] of this rank's rows (vsa_match
    records: 4 words; packed candidates: 2) on `device`.  Returns (list of
    per-rank tensors trimmed to their counts, counts).  Ragged lists are
    padded to the longest one for the collective."""
    world = dist.get_world_size()
    count = torch.tensor([local.numel() // words], dtype=torch.int64,
                         device=device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    cap = max(counts) * words
    padded = torch.zeros(max(cap, words), dtype=torch.int64, device=device)
    padded[:local.numel()] = local
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    return [g[:c * words] for g, c in zip(gathered, counts)], counts


def global_mum_filter(dist, torch, local_candidates, device, filter_fn):
    """The one exchange step of `vmatch -mum` over sharded queries.
    filter_fn(int64 tensor [n*4]) -> (number of MUMs, sum of their lengths);
    it is called on rank 0 only.  Returns (mums, sumlength, candidates) as
    job-wide totals on every rank."""
    parts, counts = all_gather_matches(dist, torch, local_candidates, device)
    nmum = sumlen = 0
    if dist.get_rank() == 0:
        allc = torch.cat(parts) if sum(counts) else torch.zeros(
            0, dtype=torch.int64, device=device)
        nmum, sumlen = filter_fn(allc)
    nmum, sumlen = all_reduce_counters(dist, torch, [nmum, sumlen], device)
    return nmum, sumlen, sum(counts)


def _no_device_rows_through_the_gather(rows):
    """the gather-everything form moves world times the rows: it is the form
    of the CPU tests and of the one-GPU rehearsal (host copies), never a
    silent fallback for rows that live on a GPU"""
    if getattr(rows, "is_cuda", False):
        raise RuntimeError(
            "vstree_amd.sharding: device rows, but the process group is not "
            "on RCCL (backend %r): the exchange would gather every rank's "
            "rows on every rank" % (rows.device,))


def _exchange_rows(dist, torch, rows, dest, device):
    """rows [N,4] int64 -> the rows every rank addressed to this rank
    (dest[i] = receiving rank), concatenated in rank order"""
    world = dist.get_world_size()
    order = torch.argsort(dest)
    rows = rows[order].contiguous()
    send = torch.bincount(dest, minlength=world).to(torch.int64)
    recv = torch.zeros_like(send)
    if is_rccl(dist):   # (decided alike on every rank)
        dist.all_to_all_single(recv, send)
        out = torch.empty((int(recv.sum().item()), MATCH_WORDS),
                          dtype=torch.int64, device=device)
        dist.all_to_all_single(out, rows, recv.tolist(), send.tolist())
        return out
    else:
        # backend without all-to-all (gloo in the CPU tests): gather all,
        # keep what is addressed to this rank
        _no_device_rows_through_the_gather(rows)
        parts, _ = all_gather_matches(dist, torch, rows.reshape(-1), device)
        dests, _ = all_gather_matches(
            dist, torch,
            torch.sort(dest).values.repeat_interleave(MATCH_WORDS), device)
        me = dist.get_rank()
        keep = [p.reshape(-1, MATCH_WORDS)[
            d.reshape(-1, MATCH_WORDS)[:, 0] == me]
            for p, d in zip(parts, dests)]
        return torch.cat(keep) if keep else rows[:0]


_META = {}
# VSA_META_STAGING=0: a fresh pageable tensor per batch (round 2's form)
_STAGED = os.environ.get("VSA_META_STAGING", "1") != "0"


def _meta_staging(torch, world, device):
    """buffers of the metadata all-gather (2*world numbers per rank), with
    numpy views of the page-locked ones"""
    key = (world, str(device))
    if key not in _META:
        n = 2 * world
        st = dict(
            host_in=torch.empty(n, dtype=torch.int64).pin_memory(),
            dev_in=torch.empty(n, dtype=torch.int64, device=device),
            dev_out=torch.empty(n * world, dtype=torch.int64, device=device),
            host_out=torch.empty(n * world, dtype=torch.int64).pin_memory())
        st["np_in"] = st["host_in"].numpy()
        st["np_out"] = st["host_out"].numpy()
        _META[key] = st
    return _META[key]


def meta_device_buffer(torch, world, device):
    """the device tensor (2*world int64) vsa_result_partition_device writes
    for partitioned_mum_filter_presorted(meta_on_device=True)"""
    return _meta_staging(torch, world, device)["dev_in"]


def partitioned_mum_filter_presorted(dist, torch, rows, send, maxright,
                                     device, filter_fn, words=MATCH_WORDS,
                                     extra=(), reduce=True, own_last=False,
                                     meta_on_device=False):
    """partitioned_mum_filter for candidates that vsa_result_partition has
    grouped by destination already (send[r] rows for rank r, maxright[r] =
    their largest right end): no sorting on this side and three collectives --
    one all-gather of the 2*world numbers of every rank (split sizes of the
    exchange and the carries), the all-to-all of the rows, the all-reduce of
    the counters.  words = 8-byte words per row: 4 for records, 2 for the
    (key, value) pairs of vsa_findmumcandidates_packed.  extra: more local
    counters to sum over the ranks in the same all-reduce; their totals
    follow the three results.  reduce=False: no all-reduce -- the counters
    returned are this rank's own, for a caller that sums them over many
    batches and reduces once at the end of the job (the only collective the
    counters need).
    own_last=True: the rows for this rank itself lie BEHIND all others
    (vsa_result_partition_own) and never enter the exchange: the all-to-all
    runs over the rows in front of them with a split of 0 for the rank
    itself, and filter_fn(own, received, carry) gets the two lists as they
    lie (vsa_mumuniqueinquery_range_packed2 takes them as one).
    meta_on_device=True (RCCL only): send and maxright are None -- the 2*world
    numbers lie in meta_device_buffer(...) already, where
    vsa_result_partition_device left them without waiting for the GPU; they
    reach the host once, gathered."""
    world, me = dist.get_world_size(), dist.get_rank()
    rows = rows.reshape(-1, words)
    # which collectives the backend has is a property of the process group,
    # decided the same way on every rank before anything is sent (a fallback
    # inside `except` would let ranks that fail for another reason issue a
    # different collective than the others)
    rccl = is_rccl(dist)
    if meta_on_device and not rccl:
        raise ValueError("meta_on_device needs the RCCL backend")
    if meta_on_device and \
            torch.cuda.current_stream() != torch.cuda.default_stream():
        # vsa_result_partition_device / vsa_findmumcandidates_grouped queue
        # their last kernels on the legacy default stream and do not wait
        # (include/vstree_amd.h): the collectives below are ordered behind
        # them only if they are issued on that stream too
        raise RuntimeError("meta_on_device: the split sizes are written on "
                           "the default stream; call this outside "
                           "torch.cuda.stream(...) contexts")
    if rccl and (_STAGED or meta_on_device):
        # one tensor in, one out, through page-locked staging buffers kept
        # from batch to batch: no allocation, no pageable copy
        st = _meta_staging(torch, world, device)
        if not meta_on_device:
            st["np_in"][:world] = send
            st["np_in"][world:] = maxright
            st["dev_in"].copy_(st["host_in"], non_blocking=True)
        dist.all_gather_into_tensor(st["dev_out"], st["dev_in"])
        st["host_out"].copy_(st["dev_out"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        table = st["np_out"].reshape(world, -1)   # [sender, 2*world]
    elif rccl:
        meta = torch.as_tensor(np.concatenate(
            [np.asarray(send, np.int64), np.asarray(maxright, np.int64)]),
            device=device)
        gathered = torch.empty(world * meta.numel(), dtype=meta.dtype,
                               device=device)
        dist.all_gather_into_tensor(gathered, meta)
        table = gathered.reshape(world, -1).cpu().numpy()
    else:
        meta = torch.as_tensor(np.concatenate(
            [np.asarray(send, np.int64), np.asarray(maxright, np.int64)]),
            device=device)
        metas = [torch.zeros_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta)
        table = torch.stack(metas).cpu().numpy()
    sends, tops = table[:, :world].copy(), table[:, world:].copy()
    # largest right end among ALL candidates of the ranges below mine
    carry = int(tops[:, :me].max()) if me > 0 else 0
    nown = int(sends[me, me]) if own_last else 0
    nrows = rows.shape[0]
    if own_last:
        # what a rank keeps does not travel
        sends[np.arange(world), np.arange(world)] = 0
    recv = [int(x) for x in sends[:, me]]
    mine = torch.empty((sum(recv), words), dtype=torch.int64, device=device)
    if not sends.any():
        # nothing travels (one rank; or every candidate lies in its own
        # rank's range): every rank sees the same table and takes this branch
        # with the others
        pass
    elif rccl:
        dist.all_to_all_single(mine, rows[:nrows - nown], recv,
                               [int(x) for x in sends[me]])
    else:
        # backend without all-to-all (gloo in the CPU tests): gather all,
        # cut out what is addressed to this rank
        _no_device_rows_through_the_gather(rows)
        parts, _ = all_gather_matches(dist, torch, rows.reshape(-1), device,
                                      words)
        keep = []
        for r, p in enumerate(parts):
            # (own_last: the diagonal of `sends` is 0 by now, which is what
            # the layout -- own rows behind the others -- needs here)
            off = int(sends[r, :me].sum())
            keep.append(p.reshape(-1, words)[off:off + recv[r]])
        mine = torch.cat(keep) if keep else rows[:0]
    if own_last:
        nmum, sumlen = filter_fn(rows[nrows - nown:].reshape(-1),
                                 mine.reshape(-1), carry)
    else:
        nmum, sumlen = filter_fn(mine.reshape(-1), carry)
    local = [nmum, sumlen, nrows] + list(extra)
    totals = (all_reduce_counters(dist, torch, local, device) if reduce
              else [int(v) for v in local])
    return tuple(totals) if extra else tuple(totals[:3])


def partitioned_mum_filter(dist, torch, local_candidates, totallength,
                           device, filter_fn):
    """`vmatch -mum` over sharded queries without a single-GPU bottleneck:
    candidates are range-partitioned by dbstart over the ranks (all-to-all),
    every rank filters its range with the carry of the ranges below it
    (vsa_mumuniqueinquery_range), counters are all-reduced.  The MUM list
    stays distributed; rank order = dbstart order.
    filter_fn(int64 tensor [n*4], carry) -> (number of MUMs, sum of lengths)
    Returns (mums, sumlength, candidates) as job-wide totals."""
    world, rank = dist.get_world_size(), dist.get_rank()
    rows = local_candidates.reshape(-1, MATCH_WORDS)
    # equal dbstarts always land on the same rank
    dest = (rows[:, 1] * world) // (int(totallength) + 1)
    mine = _exchange_rows(dist, torch, rows, dest, device)
    # carry = largest right end among all candidates of the lower ranges
    localmax = torch.zeros(1, dtype=torch.int64, device=device)
    if mine.shape[0] > 0:
        localmax[0] = (mine[:, 1] + mine[:, 0] - 1).max()
    allmax = [torch.zeros_like(localmax) for _ in range(world)]
    dist.all_gather(allmax, localmax)
    carry = max([0] + [int(m.item()) for m in allmax[:rank]])
    nmum, sumlen = filter_fn(mine.reshape(-1), carry)
    nmum, sumlen, ncand = all_reduce_counters(
        dist, torch, [nmum, sumlen, rows.shape[0]], device)
    return nmum, sumlen, ncand


def matches_to_tensor(torch, matches, device="cpu"):
    """structured numpy match array -> flat int64 tensor"""
    flat = np.ascontiguousarray(matches).view(np.uint64).astype(np.int64)
    return torch.from_numpy(flat.reshape(-1)).to(device)


def tensor_to_matches(tensor, dtype):
    a = tensor.cpu().numpy().astype(np.uint64).reshape(-1, MATCH_WORDS)
    out = np.zeros(a.shape[0], dtype)
    for i, name in enumerate(dtype.names):
        out[name] = a[:, i]
    return out


def pack_candidates(matches, lengthbits):
    """host restatement of the pair layout of vsa_findmumcandidates_packed:
    structured match array -> uint64 array [n, 2] of (key, value) rows"""
    mask = np.uint64((1 << lengthbits) - 1)
    rows = np.zeros((len(matches), 2), np.uint64)
    rows[:, 0] = (matches["dbstart"] << np.uint64(lengthbits)) | (
        mask - matches["length"])
    rows[:, 1] = (matches["queryseq"] << np.uint64(16)) | matches["querystart"]
    return rows


def unpack_candidates(rows, lengthbits, dtype):
    """(key, value) rows -> structured match array"""
    rows = np.asarray(rows, np.uint64).reshape(-1, 2)
    mask = np.uint64((1 << lengthbits) - 1)
    out = np.zeros(rows.shape[0], dtype)
    out["length"] = mask - (rows[:, 0] & mask)
    out["dbstart"] = rows[:, 0] >> np.uint64(lengthbits)
    out["queryseq"] = rows[:, 1] >> np.uint64(16)
    out["querystart"] = rows[:, 1] & np.uint64(0xFFFF)
    return out
