"""A CPU stand-in for HipShardEngine, used ONLY to drive
hashgan_amd.sharded.evaluate_shard under gloo (world_size 2, no GPU): same five
methods, NumPy + the oracle's expressions inside.  It restates the counting
plan of k_plan / k_order in the simplest possible form."""
import numpy as np
import torch
from oracle import hamming_map as O


class NumpyShardEngine:
    def __init__(self, qbits, qlab, dbbits, dblab, idx_base):
        self.q, self.ql, self.db, self.dl, self.base = qbits, qlab, dbbits, dblab, idx_base
        self.D = O.hamming_matrix(O.pack_bits(qbits), O.pack_bits(dbbits))      # [Q, Nshard]
        self.NB = qbits.shape[1] + 1

    def hist(self):
        h = np.stack([np.bincount(self.D[i], minlength=self.NB) for i in range(self.D.shape[0])])
        self.h = h.astype(np.int64)
        return torch.from_numpy(self.h.astype(np.int32))

    def plan(self, R, gathered, world, rank):
        H = gathered.numpy().astype(np.int64) if world > 1 else self.h[None]
        self.R, self.rank = R, rank
        tot = H.sum(0)                                   # [Q, NB]
        cum = np.cumsum(tot, 1)
        self.t = (cum >= R).argmax(1)
        before = H[:rank].sum(0)
        start = cum - tot                                 # global start of each bucket
        self.posbase = start + before                     # my first slot per bucket
        self.cnt_lt = np.array([start[q, self.t[q]] for q in range(tot.shape[0])])
        self.quota = R - self.cnt_lt
        self.tie_before = np.array([before[q, self.t[q]] for q in range(tot.shape[0])])

    def select_match(self):
        Q, R = self.D.shape[0], self.R
        self.idx = np.full((Q, R), 0xFFFFFFFF, dtype=np.int64)
        bits = np.zeros((Q, R), dtype=bool)
        for q in range(Q):
            order = np.argsort(self.D[q], kind="stable")
            run = {}
            ties = 0
            for n in order:
                d = self.D[q, n]
                if d < self.t[q]:
                    pos = self.posbase[q, d] + run.get(d, 0)
                    run[d] = run.get(d, 0) + 1
                elif d == self.t[q]:
                    gr = self.tie_before[q] + ties
                    ties += 1
                    if gr >= self.quota[q]:
                        continue
                    pos = self.cnt_lt[q] + gr
                else:
                    break
                self.idx[q, pos] = self.base + n
                bits[q, pos] = O.label_match(self.ql[q], self.dl[n:n + 1])[0]
        self.bits = bits
        return torch.from_numpy(np.packbits(bits, axis=1, bitorder="little"))

    def finish(self, gathered_bits, world):
        packed = gathered_bits.numpy() if world > 1 else np.packbits(self.bits, axis=1, bitorder="little")[None]
        merged = np.bitwise_or.reduce(packed, axis=0)
        m = np.unpackbits(merged, axis=1, bitorder="little")[:, :self.R].astype(bool)
        ap = np.full(m.shape[0], np.nan)
        rel = np.zeros(m.shape[0], dtype=np.int64)
        for q in range(m.shape[0]):
            a, r = O.average_precision(m[q], self.R)
            rel[q] = r
            if a is not None:
                ap[q] = a
        return ap, rel


class NumpyRankedEngine(NumpyShardEngine):
    """Adds the one-exchange form of the bet (HipShardEngine.select_ranked / merge_ranked): every shard ranks its
    own rows, the global bitmap is stitched from the gathered local ones.  The "guess" here keeps every row,
    which is a valid (if useless) superset; the merge is k_merge_ranked restated with Python loops."""

    def bet_eligible(self, R, world):
        return True

    def ranked_merge_ok(self, world):
        return True

    def sample_hist(self, R):
        return self.hist()

    def guess(self, R, gathered, world, rank):
        self.R = R

    def select_ranked(self):
        Q, R = self.D.shape[0], self.R
        bits = np.zeros((Q, R), dtype=bool)
        for q in range(Q):
            order = np.argsort(self.D[q], kind="stable")[:R]          # local rank order: distance, then index
            bits[q, :len(order)] = [O.label_match(self.ql[q], self.dl[n:n + 1])[0] for n in order]
        self.local_bits = bits
        return torch.from_numpy(self.h.astype(np.int32)), torch.from_numpy(np.packbits(bits, axis=1, bitorder="little"))

    def merge_ranked(self, gathered_hist, gathered_bits, world):
        H = gathered_hist.numpy().astype(np.int64) if world > 1 else self.h[None]
        B = gathered_bits.numpy() if world > 1 else np.packbits(self.local_bits, axis=1, bitorder="little")[None]
        Q, R = self.D.shape[0], self.R
        out = np.zeros((Q, R), dtype=bool)
        lost = False
        for q in range(Q):
            loc = [np.unpackbits(B[r, q], bitorder="little")[:R].astype(bool) for r in range(H.shape[0])]
            off = [0] * H.shape[0]
            pos = 0
            for d in range(self.NB):
                for r in range(H.shape[0]):
                    take = min(int(H[r, q, d]), R - pos)
                    out[q, pos:pos + take] = loc[r][off[r]:off[r] + take]
                    pos += take
                    off[r] += int(H[r, q, d])
                if pos >= R:
                    break
            lost = lost or pos < R
        self.bits = out
        return lost

    def finish(self, gathered_bits, world):
        if gathered_bits is None and world > 1:                       # merged already: nothing to OR
            packed = np.packbits(self.bits, axis=1, bitorder="little")[None]
            return NumpyShardEngine.finish(self, torch.from_numpy(packed), 1)
        return NumpyShardEngine.finish(self, gathered_bits, world)

    # the per-query stages split over the ranks (HipShardEngine.merge_ap_part / unpack_parts, hg_merge_ap_part)
    def merge_ap_part(self, gathered_hist, gathered_bits, world, rank):
        from hashgan_amd.sharded import shard_bounds
        lost = self.merge_ranked(gathered_hist, gathered_bits, world)            # (all queries: the stand-in is not about speed)
        ap, rel = NumpyShardEngine.finish(self, torch.from_numpy(np.packbits(self.bits, axis=1, bitorder="little")[None]), 1)
        bounds = shard_bounds(self.D.shape[0], world)
        width = max(n for _, n in bounds)
        q0, nq = bounds[rank]
        part = np.zeros((width + 1, 2), dtype=np.float64)
        part[:nq, 0], part[:nq, 1] = ap[q0:q0 + nq], rel[q0:q0 + nq]
        part[width] = (1.0 if lost else 0.0, nq)
        return torch.from_numpy(part)

    def unpack_parts(self, gathered_parts, world):
        P = gathered_parts.numpy()
        width = P.shape[1] - 1
        ns = [int(P[r, width, 1]) for r in range(world)]
        ap = np.concatenate([P[r, :ns[r], 0] for r in range(world)])
        rel = np.concatenate([P[r, :ns[r], 1] for r in range(world)]).astype(np.int64)
        return ap, rel, bool(P[:, width, 0].any())


class NumpyRoutedEngine(NumpyRankedEngine):
    """The same bet with its exchanges routed by query owner (HipShardEngine.pack_sample_by_owner ... merge_ap_owned over
    comm.all_to_all): every table is cut into one block per owner of its queries, block o holding rows q0(o) .. of them, and
    the owner stitches and evaluates only what it received.  Blocks are [world, ...] tensors, like the HIP engine's."""

    def _owners(self, world):
        from hashgan_amd.sharded import shard_bounds
        b = shard_bounds(self.D.shape[0], world)
        return b, max(n for _, n in b)

    def sample_hist(self, R):
        self.R = R
        return self.hist()

    def pack_sample_by_owner(self, world):
        bounds, width = self._owners(world)
        out = np.zeros((world, width, self.NB), np.int32)
        for o, (q0, nq) in enumerate(bounds):
            out[o, :nq] = self.h[q0:q0 + nq]
        return torch.from_numpy(out)

    def guess_owned(self, R, received, world, rank):
        # (the stand-in's "guess" keeps every row: the answers carry nothing, but they make the second all-to-all happen)
        _, width = self._owners(world)
        assert tuple(received.shape) == (world, width, self.NB)
        return torch.zeros((world, width, 4), dtype=torch.int32)

    def guess_finish(self, R, answers, world, rank):
        _, width = self._owners(world)
        assert tuple(answers.shape) == (world, width, 4)
        self.R = R

    def pack_ranked_by_owner(self, world):
        bounds, width = self._owners(world)
        nbyte = (self.R + 7) // 8
        packed = np.packbits(self.local_bits, axis=1, bitorder="little")
        out = np.zeros((world, width, self.NB * 4 + nbyte), np.uint8)
        for o, (q0, nq) in enumerate(bounds):
            out[o, :nq, :self.NB * 4] = self.h[q0:q0 + nq].astype(np.int32).view(np.uint8).reshape(nq, self.NB * 4)
            out[o, :nq, self.NB * 4:] = packed[q0:q0 + nq]
        return torch.from_numpy(out)

    def merge_ap_owned(self, received, world, rank):
        bounds, width = self._owners(world)
        q0, nq = bounds[rank]
        blk = received.numpy()
        R = self.R
        part = np.zeros((width + 1, 2), dtype=np.float64)
        lost = False
        for i in range(nq):
            H = np.stack([blk[r, i, :self.NB * 4].copy().view(np.int32) for r in range(world)]).astype(np.int64)     # [world, NB]
            loc = [np.unpackbits(blk[r, i, self.NB * 4:], bitorder="little")[:R].astype(bool) for r in range(world)]
            out = np.zeros(R, dtype=bool)
            off = [0] * world
            pos = 0
            for d in range(self.NB):
                for r in range(world):
                    take = min(int(H[r, d]), R - pos)
                    out[pos:pos + take] = loc[r][off[r]:off[r] + take]
                    pos += take
                    off[r] += int(H[r, d])
                if pos >= R:
                    break
            lost = lost or pos < R
            a, rel = O.average_precision(out, R)
            part[i] = (np.nan if a is None else a, rel)
        part[width] = (1.0 if lost else 0.0, nq)
        return torch.from_numpy(part)
