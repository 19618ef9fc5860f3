/*
 * hashgan_amd -- C ABI of the MI355X-native retrieval-evaluation path.
 *
 * The reference has no FFI: its boundary for this path is the Python call
 *     MAPs(R).get_maps_by_feature(database, query)      lib/metric.py:4-24
 * made from evaluate()                                   main.py:161-164.
 * hashgan_amd/metric.py keeps that call surface; everything below is what that
 * Python binds through ctypes.  Conventions:
 *   - extern "C", plain pointers and sizes, no C++/torch types;
 *   - every function returns 0 (HG_OK) or a negative HG_ERR_* code and never
 *     throws; hg_last_error() gives the message of the calling thread's last
 *     failure;
 *   - "host" pointers are caller-owned host memory, "dev" pointers are device
 *     addresses (hipMalloc'ed by the caller or by torch) on the context's GPU;
 *   - a context owns one GPU stream; every call is complete (stream
 *     synchronised) when it returns; a context is not thread safe.
 *
 * Data layout (pinned by tests/test_capi.py::test_pack_sign_matches_python_packing and
 * tests/test_oracle_golden.py::test_pack_bits_layout):
 *   codes   uint64 [n][W], W = ceil(b/64); bit j of a code is bit (j % 64) of
 *           word j / 64; bit value = (feature[j] > 0); pad bits are zero.
 *   labels  uint64 [n][LW], LW = ceil(C/64), bit c set <=> label[c] != 0.
 *   ranked lists: position k of query q is element [q*R + k].
 *
 * Canonical order: Hamming distance ascending, then database
 * index ascending -- what np.argsort(-ips, 1) (metric.py:14) yields for +-1
 * codes once ties are broken by index.
 */
#ifndef HASHGAN_AMD_H
#define HASHGAN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_OK 0
#define HG_ERR_ARG (-1)    /* bad argument (R > N, b out of range, null pointer ...) */
#define HG_ERR_HIP (-2)    /* a HIP runtime call or kernel failed */
#define HG_ERR_STATE (-3)  /* call sequence violated (e.g. hg_select before hg_plan) */
#define HG_ERR_NOMEM (-4)  /* device allocation failed */

#define HG_MAX_BITS 255        /* longest supported code: a Hamming distance has to fit the 8 bits it gets in
                                  the records and in the ranked lists (two 256-bit codes can be 256 apart) */
#define HG_IDX_NONE 0xFFFFFFFFu /* ranked-list slot owned by another shard */

typedef struct hg_ctx hg_ctx;

const char* hg_last_error(void);
int hg_version(void);
int hg_device_count(int* count);

/* One context per (process, GPU).  Replaces nothing in the reference (it runs
 * the metric on the host); it is the handle the Python MAPs object keeps. */
int hg_init(int device, hg_ctx** out);
int hg_destroy(hg_ctx* ctx);

/* Binarise-and-pack, host side: bit j = (x[i*b + j] > 0).  This is the sign()
 * the reference never applies (its tanh features go straight into np.dot,
 * lib/architecture.py:147 -> metric.py:13); for +-1 features it is exact. */
int hg_pack_sign_f32(const float* host_x, int64_t n, int b, uint64_t* host_out);

/* database.output / database.label of metric.py:13,19 as packed codes/labels.
 * idx_base is the global index of this shard's row 0, n_total the size of the
 * whole (all-shard) database; single GPU: idx_base = 0, n_total = N. */
int hg_set_database(hg_ctx* ctx, const uint64_t* host_codes, const uint64_t* host_labels,
                    int64_t N, int b, int C, int64_t idx_base, int64_t n_total);
/* query.output / query.label of metric.py:13,17. Same b and C as the database. */
int hg_set_queries(hg_ctx* ctx, const uint64_t* host_codes, const uint64_t* host_labels, int64_t Q);

/* The same two calls fed with what forward_all() (main.py:151-158) actually returns: float32
 * features [n][b] and int64 labels [n][C].  Binarise (bit = x > 0) and pack run on a pool of host threads BEFORE the
 * upload (16 MB instead of 339 MB cross PCIe at C2; option "host_pack" = 0: upload raw, pack on the GPU).  The float
 * table itself follows only if it will be ranked by inner product: option "keep_floats" = 2 (default) uploads it iff the
 * database is not a +-1 code, 1 always, 0 never; the queries' floats follow the database's.
 * *bad_codes counts feature entries outside {-1, 0, +1}, *bad_labels label entries outside
 * {0, 1}: the caller decides what non-binary features mean (the Python mirror ranks them by inner product
 * like metric.py:13-14 unless binarize=True); hg_get_stat has the finer census (zeros, minus ones). */
int hg_set_database_f32(hg_ctx* ctx, const float* host_features, const int64_t* host_labels, int64_t N, int b, int C,
                        int64_t idx_base, int64_t n_total, int64_t* bad_codes, int64_t* bad_labels);
int hg_set_queries_f32(hg_ctx* ctx, const float* host_features, const int64_t* host_labels, int64_t Q,
                       int64_t* bad_codes, int64_t* bad_labels);
/* Packed device tables back to the host: which = 0 database, 1 queries; codes as dense
 * uint32 [n][ceil(b/32)], labels uint64 [n][ceil(C/64)]. */
int hg_get_packed(hg_ctx* ctx, int which, uint32_t* host_codes, uint64_t* host_labels);

/* ---- staged pipeline (what multi-GPU orchestration drives) ------------------
 * hg_hist    metric.py:13   XOR+popcount of every (query, db row) pair, reduced
 *                           to per-query distance histograms of this shard.
 * hg_plan    metric.py:14   per query: threshold distance t (the R-th smallest
 *                           over ALL shards), tie quota, output offsets.
 *                           dev_hist_all = G shard histograms, each laid out as
 *                           hg_hist_buffer describes, shard-rank order; pass
 *                           NULL, G = 1, rank = 0 on a single GPU.
 * hg_select  metric.py:14 + [0:R] at :19   second pass over the pairs: emits the
 *                           shard's members of the global top-R into their
 *                           global rank positions, canonical order, together
 *                           with their label-match bits (metric.py:17-19).
 * hg_match   metric.py:17-19  label match -> bit rows.  Already done by
 *                           hg_select for C <= 128 classes (then a no-op);
 *                           a gather pass over the ranked lists otherwise.
 * hg_merge_match            OR of G shards' bit rows (after an all-gather).
 * hg_ap      metric.py:20-23  per-query AP in float64, same rounding and
 *                           summation order as NumPy (pairwise, 8192 chunks).
 */
int hg_hist(hg_ctx* ctx);
/* The exchange unit of every histogram stage: uint32 [b+1][Qpad] followed by 64 tail words
 * ([0] "a slice overflowed" flag, [1] rows the pass visited). */
int hg_hist_buffer(hg_ctx* ctx, void** dev_ptr, int64_t* nbytes);
int hg_plan(hg_ctx* ctx, int64_t R, const uint32_t* dev_hist_all, int G, int rank);
int hg_select(hg_ctx* ctx);
int hg_match(hg_ctx* ctx);
int hg_match_buffer(hg_ctx* ctx, void** dev_ptr, int64_t* nbytes); /* uint64 [Q][ceil(R/64)] */
int hg_merge_match(hg_ctx* ctx, const uint64_t* dev_bits_all, int G);
int hg_ap(hg_ctx* ctx);

/* ---- staged optimistic sequence (R << N): one pass over the pairs instead of two ----
 * hg_bet_eligible      same verdict on every rank (uses only R, n_total, world, options)
 * hg_sample_hist       histogram of every k-th row batch            -> hg_hist_buffer -> all-gather
 * hg_guess             threshold guess from the G sample histograms ("guess_sigma" = 5 standard deviations high)
 * hg_select_candidates ONE pass over the pairs: superset of the members as records, plus the
 *                      exact histogram of those records              -> hg_hist_buffer -> all-gather
 * hg_rank              exact plan from the G record histograms + ordering + match bits.
 *                      *bet_lost = 1 (on every rank alike) when the superset was short of R rows
 *                      or a slice overflowed anywhere: run hg_hist/hg_plan/hg_select instead. */
int hg_bet_eligible(hg_ctx* ctx, int64_t R, int world, int* eligible);
int hg_sample_hist(hg_ctx* ctx, int64_t R);
int hg_guess(hg_ctx* ctx, int64_t R, const uint32_t* dev_hist_all, int G, int rank);
int hg_select_candidates(hg_ctx* ctx);
int hg_rank(hg_ctx* ctx, const uint32_t* dev_hist_all, int G, int rank, int* bet_lost);
/* With option "defer_verdict" = 1 hg_rank does not wait for the verdict (*bet_lost = -1): the caller carries on as
 * if the bet held -- match bits, their exchange, hg_ap -- and asks here once, next to its final download; on
 * *bet_lost = 1 it discards those results and runs the exact sequence.  Removes the only host round trip from
 * the middle of a sharded step. */
int hg_bet_verdict(hg_ctx* ctx, int* bet_lost);
/* The bet with ONE record pass and ONE exchange after the guess (AP only, no ranked lists):
 * hg_select_ranked   select with the shared guess, then rank this shard's own records.  Leaves the shard's
 *                    per-distance record counts in hg_hist_buffer and its match bitmap in LOCAL rank order
 *                    (metric.py:14,17-19 restricted to the shard) in hg_match_buffer -> all-gather both
 * hg_merge_ranked    shards own contiguous index ranges, so the global order is, per distance, shard 0's rows,
 *                    then shard 1's, ...: derives the cut from the gathered counts (metric.py:19's [0:R]) and
 *                    stitches the global match bitmap from bit ranges of the local ones; then hg_ap.
 *                    *bet_lost as in hg_rank (-1 with "defer_verdict"); G <= 64. */
int hg_select_ranked(hg_ctx* ctx);
int hg_merge_ranked(hg_ctx* ctx, const uint32_t* dev_hist_all, const uint64_t* dev_bits_all, int G, int* bet_lost);
/* The same with the per-query stages split over the ranks (hashgan_amd/sharded.py::evaluate_shard): after the all-gather
 * of record counts and local bitmaps, rank r merges and evaluates (k_merge_ranked + k_ap) only its own contiguous share
 * [q0, q0 + nq) of the queries instead of all Q on every rank.  Its results leave in one device buffer *dev_part of
 * (width + 1) pairs of doubles -- pair i < nq = {AP, hit count} of query q0 + i, pairs nq .. width - 1 zero, pair `width` =
 * {this rank's lost-bet flag, nq}; width >= nq is the same on every rank so that the parts can be all-gathered.
 * hg_unpack_parts takes the gathered G parts (rank order = query order), fills AP / hit counts of all Q queries on the host
 * and returns the bet's verdict (the OR of the ranks' flags: computed from gathered data, identical on every rank), doing
 * hg_bet_verdict's bookkeeping.  The reference has no counterpart (lib/metric.py:16-24 loops over all queries in one process). */
int hg_merge_ap_part(hg_ctx* ctx, const uint32_t* dev_hist_all, const uint64_t* dev_bits_all, int G, int64_t q0, int64_t nq,
                     int64_t width, void** dev_part, int64_t* nbytes);
int hg_unpack_parts(hg_ctx* ctx, const void* dev_parts_all, int G, int64_t width, double* host_ap, int64_t* host_rel, int* bet_lost);
/* The same sharded bet with its exchanges ROUTED BY QUERY OWNER (round 4).  The queries are split over the ranks like the
 * per-query stages above (contiguous, near-equal shares: rank o owns [q0(o), q0(o) + nq(o)), width = the largest share);
 * what a stage needs of a query it needs from every shard but only on the query's owner, so the tables travel by
 * all-to-all (hg_alltoall), one block per destination, instead of every rank receiving every query's rows of every
 * shard: at C4 on 8 GPUs 10.6 MB of ingress per GPU and step instead of 80 MB.  Results are those of the all-gather form
 * (and of one GPU) bit for bit.
 *   hg_sample_hist -> hg_pack_sample_by_owner   *dev_ptr = [G] blocks {rows sampled, counts [planes][width]}      -> all-to-all
 *   hg_guess_owned(dev_recv)                    the owner's guess of ITS queries; *dev_ptr = [G] answers [width][4 x u32]
 *                                               {T, sampled rows ahead of shard r's own, sample count to reach, found}  -> all-to-all
 *   hg_guess_finish(dev_answers)                the shared cut T + this shard's sstar for ALL queries; budget as hg_guess
 *   hg_select_ranked -> hg_pack_ranked_by_owner *dev_ptr = [G] blocks {record counts [b+1][width], tail, local bitmaps [width][RW]} -> all-to-all
 *   hg_merge_ap_owned(dev_recv)                 hg_merge_ap_part on the received blocks -> this rank's part -> hg_allgather -> hg_unpack_parts */
int hg_pack_sample_by_owner(hg_ctx* ctx, int G, void** dev_ptr, int64_t* nbytes_per_peer);
int hg_guess_owned(hg_ctx* ctx, int64_t R, const void* dev_recv, int G, int rank, void** dev_ptr, int64_t* nbytes_per_peer);
int hg_guess_finish(hg_ctx* ctx, int64_t R, const void* dev_answers, int G, int rank);
int hg_pack_ranked_by_owner(hg_ctx* ctx, int G, void** dev_ptr, int64_t* nbytes_per_peer);
int hg_merge_ap_owned(hg_ctx* ctx, const void* dev_recv, int G, int rank, void** dev_part, int64_t* nbytes);

/* Ranked lists in global-position space: uint32 idx [Q][R] (HG_IDX_NONE where a
 * slot belongs to another shard), uint8 dist [Q][R] (0xFF there). */
int hg_topr_buffers(hg_ctx* ctx, void** dev_idx, void** dev_dist, int64_t* n_slots);
/* Element-wise merge of G shards' ranked lists after an all-gather (min picks
 * the one owner of every slot). dev_idx_all: [G][Q][R], dev_dist_all likewise. */
int hg_merge_topr(hg_ctx* ctx, const uint32_t* dev_idx_all, const uint8_t* dev_dist_all, int G);

/* ---- one-shot forms (single shard) --------------------------------------------
 * Every stage is enqueued back to back with one synchronisation.  When R << N
 * they replace the full histogram pass by a sampled one and a verified guess of
 * the threshold (identical results; falls back to the staged sequence if the
 * verification fails).  hg_map skips the idx/dist lists -- mAP only needs the
 * match bits; use hg_topr when the ranked lists themselves are wanted. */
int hg_topr(hg_ctx* ctx, int64_t R);
int hg_map(hg_ctx* ctx, int64_t R, double* host_ap, int64_t* host_rel);

/* ---- real-valued features (what main.py feeds when nothing is binarised) ----
 * Ranking by float32 inner product, lib/metric.py:13-14 as written, on the float tables kept by
 * hg_set_database_f32 / hg_set_queries_f32 (up to 255 features; one context holds the whole table -- several GPUs split
 * the QUERIES, hashgan_amd/sharded.py::evaluate_real_queries).  Order: inner product descending,
 * database index ascending.  The product's summation order is fixed (one float32 fma chain in feature
 * order: what the chained v_mfma_f32_32x32x2_f32 computes) and restated exactly by oracle/real_map.py; it equals the reference's
 * np.dot wherever float32 rounding does not reorder near-equal products, exactly so on inputs
 * whose arithmetic is exact.  hg_map_real = ranking + label match + AP; hg_topr_real = ranking only. */
int hg_map_real(hg_ctx* ctx, int64_t R, double* host_ap, int64_t* host_rel);
int hg_topr_real(hg_ctx* ctx, int64_t R);
int hg_get_topr_real(hg_ctx* ctx, uint32_t* host_idx, float* host_scores);   /* [Q][R] each */

/* ---- results to the host ----------------------------------------------------- */
int hg_get_topr(hg_ctx* ctx, uint32_t* host_idx, uint8_t* host_dist);   /* [Q][R] each */
int hg_get_match(hg_ctx* ctx, uint8_t* host_imatch);                    /* [Q][R] of 0/1 */
int hg_get_ap(hg_ctx* ctx, double* host_ap, int64_t* host_rel);         /* [Q]; ap = NaN where rel == 0 */
int hg_get_hist(hg_ctx* ctx, uint32_t* host_hist);                      /* [b+1][Q] of this shard */

/* ---- collectives: RCCL over xGMI, one process per GPU (the reference has no counterpart --
 * main.py:260-263 only sets CUDA_VISIBLE_DEVICES) --------------------------------------------------------
 * librccl.so.1 is dlopen'ed by the first hg_comm_* call (HG_RCCL_LIBRARY overrides the search); a single-GPU
 * process never loads it.  Rank 0 obtains a unique id and hands its HG_COMM_ID_BYTES bytes to the other ranks out
 * of band (hashgan_amd/sharded.py: a file next to the launcher's rendezvous), then every rank calls hg_comm_init.
 * Collectives run on the context's own stream, ordered with its kernels; with "stage_sync" = 1 (default) a
 * call is complete when it returns.
 *   hg_allgather        every rank contributes nbytes from dev_src; *dev_gathered = [world][nbytes] in a buffer the
 *                       context owns (slot 0..3: that many gathered buffers can be live at once), rank order.
 *                       This is the exchange of every staged sequence above (histograms, match bitmaps).
 *   hg_allgather_topr   the north star's exchange: the shards' ranked (idx, dist) lists all-gathered and merged
 *                       (hg_merge_topr); hg_get_topr then returns the global lists on every rank.
 *   hg_allreduce_max_f64 / hg_barrier   benchmark plumbing (max of the ranks' step times; barrier). */
#define HG_COMM_ID_BYTES 128
int hg_comm_unique_id(uint8_t* id);
int hg_comm_init(hg_ctx* ctx, const uint8_t* id, int rank, int world);
int hg_comm_destroy(hg_ctx* ctx);
int hg_comm_info(hg_ctx* ctx, int* rank, int* world);          /* *world = 0: no communicator */
int hg_allgather(hg_ctx* ctx, int slot, const void* dev_src, int64_t nbytes, void** dev_gathered);
/*   hg_alltoall         dev_src = [world][nbytes_per_peer]: block r goes to rank r; *dev_out = [world][nbytes_per_peer], block r
 *                       = what rank r sent here (a context buffer, slot 0..3 shared with hg_allgather).  The exchange of
 *                       the owner-routed sequence below. */
int hg_alltoall(hg_ctx* ctx, int slot, const void* dev_src, int64_t nbytes_per_peer, void** dev_out);
int hg_allgather_topr(hg_ctx* ctx);
int hg_allreduce_max_f64(hg_ctx* ctx, double* host_inout);
int hg_barrier(hg_ctx* ctx);
/* Context-owned device scratch (slot 0..3, grows only) and a stream-ordered device-to-device copy: what an
 * in-process communicator needs to do hg_allgather's job between several contexts of ONE process
 * (virtual shards on one GPU: tests/test_sharded_gpu.py). */
int hg_scratch(hg_ctx* ctx, int slot, int64_t nbytes, void** dev_ptr);
int hg_memcpy_dtod(hg_ctx* ctx, void* dev_dst, const void* dev_src, int64_t nbytes);
/* The same between a device address and host memory, complete on return (a communicator that goes through the host,
 * for multi-process dry runs on ONE GPU: tests/file_comm.py). */
int hg_memcpy_dtoh(hg_ctx* ctx, void* host_dst, const void* dev_src, int64_t nbytes);
int hg_memcpy_htod(hg_ctx* ctx, void* dev_dst, const void* host_src, int64_t nbytes);

/* Wait until everything the context has enqueued is done (with "stage_sync" = 0 nothing else does). */
int hg_synchronize(hg_ctx* ctx);

/* Run the context on a stream of the caller's (NULL: back to a private one).  With option
 * "stage_sync" = 0 the staged calls (hg_allgather included) only enqueue: every stage and the collectives
 * between them then sit on ONE stream with no host synchronisation except the bet's verdict and the
 * final hg_get_*. */
int hg_set_stream(hg_ctx* ctx, void* hip_stream);

/* ---- tuning and measurement -------------------------------------------------- */
/* key: "stage_sync" (see hg_set_stream), "target_units" (wavefront-sized units the pair passes are split into),
 * "min_segment" (rows), "max_segments", "optimistic" (0/1: one-shot calls may bet on a sampled
 * threshold -- verified on device, exact fallback), "sample_stride" (0 = auto),
 * "guess_sigma", "staged_lists" (0/1: hg_select materialises idx/dist lists),
 * "cand_budget_x10" (record budget of the bet per query, tenths of R), "rank_waves" (0 = auto, 4, 16),
 * "select_mfma" (1: the bet's select pass runs on the matrix cores -- fp4 MFMA distance tiles,
 * k_select_mx; 0: vector-ALU xor+popcount k_select; same records either way), "probe_select" (probe build only --
 * python -m hashgan_amd.build --probes: bits 2/4/8 switch parts of the matrix-core kernels' drain off; the bet
 * then fails and the exact sequence runs, so results stay right; the production library refuses the key),
 * "rank_direct" (1, default: R = N on one shard is ranked straight from the packed tables by one counting-sort kernel,
 * k_rank_direct, when its LDS fits; 2: also N/8 < R < N; 0: k_rank_fused's direct mode), "rank_direct_lds" (80: KB of LDS per block),
 * "rank_wave" (40, default: the bet's rank stage runs one wavefront per query, k_rank_wave, when a query's list of one-byte
 * records is short -- capacity value/10 x the shard's share of R + 256 records of LDS per query, used when that is at most
 * "rank_wave_max" = 4608 records: a sharded rank, a small R; 0 = always the block-per-query k_rank_cnt),
 * "select_packed" (several rows per MFMA accumulator: 3, default = k_select_mx3 (three rows through per-row MX scales,
 * batched drain) for codes of <= 64 bits with one-byte records, k_select_mx2 (two rows) for <= 32 bits otherwise; 1 =
 * k_select_mx2 for <= 32 bits only, 2 = k_select_mx2 up to 64 bits, 0 = never), "rank_lds" (0/1), "rank_cnt" (0/1:
 * the bet's rank stage as a per-thread counting sort, k_rank_cnt),
 * "host_pack", "keep_floats", "pack_threads" (hg_set_*_f32, see there), "second_bet" (1, default: a one-shot bet that too many queries lost is retried once with twice the margin and record
 * budget before the exact two-pass sequence runs; if that loses too the slices are widened -- "cap_boost" x8, then x64, kept for
 * the next calls on this database: hits crowded into few segments, e.g. rows stored class by class),
 * "cap_boost" (1..4096: multiplier on the slices' record budget; every database load sets it back to 1; the sharded
 * sequence raises it on all ranks alike after a lost bet, sharded.HipShardEngine.widen_slices), "compact_records" (1, default: when no ranked lists are wanted the matrix-core select writes one-byte records
 * {match, dist} through per-slice LDS rings instead of 8-byte {idx, dist, match} records),
 * "hist_mfma" (histogram passes -- the sampled pass of the bet, the full pass of the one-shot exact sequence: 2, default:
 * the integer matrix instruction delivers the counter addresses, k_hist_i8, codes of <= 128 bits; 1: fp4 distances,
 * k_hist_mx; 0: vector ALU), "exact_mfma" (1, default: the one-shot exact sequence selects on the matrix cores when
 * R << N), "ap_recip" (1, default: the AP kernel replaces its division by three multiply-adds against a table of
 * correctly rounded reciprocals of the ranks -- the same bits; 0: divide), "lds_pad" (extra LDS per block of the
 * matrix-core select: occupancy experiments),
 * "real_mfma" (real-valued select pass -- 2, default: bfloat16 matrix-core filter with a rigorous margin, then the exact
 * float32 chain for the rows it keeps; 1: every pair exactly on the float32 matrix-core instruction; 0: vector ALU; same
 * lists either way), "real_sort_lds" (1, default: after the filter a query's records are ranked by one LDS-resident
 * kernel when they fit; 0: always the global-memory radix passes), "real_groups" (1, default: without a cut -- every row a
 * record, R = N -- the rows are split by score range into LDS-sized groups and ordered group by group; the exact float32
 * pair pass then stands in for filter + rescore; 0: the radix passes),
 * "real_queries_per_lane", "real_segment_bytes" (real-valued path), "step_graph" (0, default: always enqueue kernel
 * by kernel; 1: hg_map captures its one-shot sequence into a hipGraph the second time it sees the same problem and
 * replays it afterwards). */
int hg_set_option(hg_ctx* ctx, const char* key, int64_t value);
/* key: "optimistic_runs", "optimistic_fallbacks" (all queries rerun exactly), "optimistic_requeried"
 * (single queries rerun exactly after losing their bet), "optimistic_rebets" (second and widened bets), "cap_boost", "real_cap_boost" (the same for hg_map_real), "real_grouped" (the last real-valued ranking ordered lists beyond the LDS group by group), "last_optimistic", "device_bytes", "segments",
 * "segment_rows", "slice_capacity", "record_row"; census of the float tables loaded by hg_set_*_f32 --
 * "db_nonbinary" / "q_nonbinary" (entries outside {-1,0,+1}), "db_zeros" / "q_zeros", "db_minus_ones" /
 * "q_minus_ones" -- from which the caller tells +-1 codes, {0,1} bits and real-valued features apart;
 * "db_floats" / "q_floats" (the float tables are on the GPU), "probe_build", "graph_captures", "graph_replays";
 * of the last real-valued ranking: "real_attempts" (1 = the first sampled cut held), "real_filtered" (filter + rescore
 * ran), "real_lds_ranked" (the LDS-resident rank kernel produced the lists). */
int hg_get_stat(hg_ctx* ctx, const char* key, int64_t* value);
/* Work buffers only grow; hg_trim frees everything except the resident code/label/feature tables
 * (stat "device_bytes" reports what the context holds). */
int hg_trim(hg_ctx* ctx);
/* HIP-event timing of the kernels launched on the context's stream.  on = 2: every kernel; 1: only the
 * select pass over the query x database pairs (k_select, k_select_mx*: the roofline kernel) -- two events per launch keep
 * consecutive kernels from being dispatched back to back, ~4 us each; 0: off.  Levels 1 and 2 also time the
 * whole one-shot step on the GPU ("step_gpu_span": first enqueue to the last byte of the result download), so
 * wall time per step - step_gpu_span = what the host adds.  Inside a captured step the events are event-record
 * nodes of the graph.  Option "timing_every" = n (default 1) makes level 1 record on every n-th one-shot step only:
 * the events of a step cost it ~0.025 ms (1.04 -> 1.07 ms at 10k x 1M), sampling one step in four keeps the average
 * kernel duration live at a quarter of that. */
int hg_timing_enable(hg_ctx* ctx, int on);
int hg_timing_reset(hg_ctx* ctx);
/* Fills up to cap entries; name[i] points to static strings. Returns count via *n. */
int hg_timing_read(hg_ctx* ctx, int cap, const char** names, double* total_ms, int64_t* launches, int* n);

#ifdef __cplusplus
}
#endif
#endif /* HASHGAN_AMD_H */
