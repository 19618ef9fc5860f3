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
 * like metric.py:13-14 unless binarize=True); hg_get_census has the finer census (zeros, minus ones). */
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
/* hg_map in two halves, for a caller that evaluates batch after batch (lib/metric.py:4-24 once per batch of queries):
 * hg_map_begin enqueues a step -- its kernels and the download of verdict, APs and hit counts into a pinned block of its
 * own -- and returns; hg_map_end waits for the OLDEST step in flight and hands over what hg_map would have returned.  Up to
 * two steps may be in flight (a third hg_map_begin is HG_ERR_STATE), so the GPU starts one step the moment the previous one
 * ends.  A step is only enqueued blind when the last synchronous hg_map on the same tables, options and R won its bet
 * outright; otherwise hg_map_begin runs the whole call itself.  A blind step that loses its bet is run again by hg_map_end.
 * Replacing the tables between the two halves is allowed only for steps that won: hg_map_end on a LOST step whose query or database
 * table has been loaded again since its hg_map_begin (whatever the new table's size) returns HG_ERR_STATE -- never the new batch's
 * results under the old step's name.  A new query table of the SAME size on an unchanged database and configuration keeps the licence
 * to enqueue blind, and hg_set_queries does not wait for the stream (the tables travel through a pinned block of the context's own):
 * a caller that hands over batch after batch -- hg_set_queries, hg_map_begin, the previous batch's hg_map_end -- keeps two steps in
 * flight.  Stats "map_async_steps" / "map_async_redone".  Test hook: option "handicap_next_bet" (0..64; not a configuration change)
 * puts the NEXT bet's guess that many deviations below the expected count, once: the bet loses. */
int hg_map_begin(hg_ctx* ctx, int64_t R);
int hg_map_end(hg_ctx* ctx, double* host_ap, int64_t* host_rel);

/* ---- real-valued features (what main.py feeds when nothing is binarised) ----
 * Ranking by float32 inner product, lib/metric.py:13-14 as written, on the float tables kept by
 * hg_set_database_f32 / hg_set_queries_f32 (up to 255 features; one context holds the whole table -- several GPUs split
 * the QUERIES, hashgan_amd/sharded.py::evaluate_real_queries).  Order: inner product descending,
 * database index ascending.  The product's summation order is fixed (one float32 fma chain in feature
 * order: what the chained v_mfma_f32_32x32x2_f32 computes) and restated exactly by oracle/real_map.py; it equals the reference's
 * np.dot wherever float32 rounding does not reorder near-equal products, exactly so on inputs
 * whose arithmetic is exact.  hg_map_real = ranking + label match + AP; hg_topr_real = ranking only.
 * hg_get_topr_real copies the ranked lists of the last hg_topr_real; hg_map_real leaves them unwritten where it ranks in LDS (as hg_map
 * does for codes: the lists are Q x R x 8 bytes of stores the APs do not need) -- HG_ERR_STATE then, unless option "real_map_lists" is 1. */
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
/* One rank's WHOLE step of the database-sharded bet (AP only, exchanges routed by query owner), enqueued back to back on the
 * context's stream with a single synchronisation at its final download -- the sequence hg_sample_hist -> hg_pack_sample_by_owner ->
 * hg_alltoall -> hg_guess_owned -> hg_alltoall -> hg_guess_finish -> hg_select_ranked -> hg_pack_ranked_by_owner -> hg_alltoall ->
 * hg_merge_ap_owned -> hg_allgather -> hg_unpack_parts in one call.  host_ap / host_rel: [Q] of ALL queries, identical on every rank.
 * *bet_lost: 0 the bet held; 1 lost on some shard (the same on every rank: widen the slices -- "cap_boost" -- and call again, or run
 * the staged exact sequence); -1 nothing was enqueued (hg_bet_eligible says no, or more shards than hg_merge_ranked takes).
 * The exchange is the context's communicator (hg_comm_init).  Without one, replica_world >= 1 makes every peer a replica of this
 * rank (device-to-device copies of its own blocks): what ONE rank of a replica_world-GPU run executes, for timing on one GPU;
 * replica_world = 1 is the one-GPU result. */
int hg_shard_step(hg_ctx* ctx, int64_t R, int replica_world, double* host_ap, int64_t* host_rel, int* bet_lost);
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
/* Engine options (33 keys; DESIGN.md section 9 lists every key with its default and the test that sets it):
 *   geometry      "target_units" (16384: wavefront-sized units the pair passes are split into), "min_segment" (256 rows), "max_segments" (2048)
 *   the bet       "optimistic" (1: one-shot calls bet on a sampled threshold -- verified on the device, exact fallback), "sample_stride" (0 = auto: every
 *                 24th tile of 16 rows), "guess_sigma" (5: margin of the guess in standard deviations of the sampled count), "cand_budget_x10" (40: record
 *                 budget per query in tenths of R), "second_bet" (1: a lost bet is retried once with twice the margin before the exact sequence),
 *                 "cap_boost" (1..4096: multiplier of the slices' capacity; every database load sets it back to 1, lost bets raise it; raising it also
 *                 forgives the sharded bet just lost), "crowd_probe" (1: the first bet on a database measures how its near rows crowd and widens the slices)
 *   kernels       "select_mfma" (1: the pair passes of bet and exact sequence on the matrix cores; 0: vector-ALU xor + popcount -- same records),
 *                 "select_packed" (3: k_select_mx3 / k_select_mx4, several distances per accumulator, one-byte records; else k_select_mx),
 *                 "compact_records" (1: one-byte records {match, dist} when no ranked lists are wanted), "hist_mfma" (2: k_hist_i8; 1: k_hist_mx; 0: k_hist),
 *                 "rank_lds" (2: k_rank_lean where it applies, else k_rank_cnt; 1: k_rank_cnt; 0: k_rank_fused only), "rank_slices" (7000: smallest R whose
 *                 bet is ranked by k_rank_dense<slices>; 0: off), "rank_dense" (1: N/8 < R <= N through the byte matrix), "rank_dense_gbm" (-1: auto; 1 / 0:
 *                 its bitmap in global memory / LDS), "dense_budget_mb" (16384), "all_rows_shortcut" (1: R = N needs no histogram and no plan),
 *                 "fuse_ap" (1: the AP leaves from the rank kernel's epilogue), "inline_leftovers" (1: queries the rank kernel declined are ranked within
 *                 the next step's stream), "ap_recip" (1: the AP's division as three multiply-adds against correctly rounded reciprocals -- the same bits), "ap_wide" (1: k_ap with 512 threads per query when the queries are few and their lists long)
 *   staging       "stage_sync" (1; 0: staged calls and collectives only enqueue, see hg_set_stream), "defer_verdict" (0; 1: hg_rank does not wait for the
 *                 bet's verdict), "staged_lists" (1: the staged hg_select materialises the idx / dist lists), "step_graph" (0; 1: hg_map replays its
 *                 sequence as a hipGraph), "timing_every" (1: kernel timing brackets every n-th step)
 *   loading       "host_pack" (1: hg_set_*_f32 pack on the host's cores before the upload), "keep_floats" (hg_set_*_f32: 0 never / 1 always / 2 = only
 *                 if not a +-1 code: keep the float table on the device)
 *   real-valued   "real_mfma" (2: bf16 matrix-core filter + exact float32 rescoring; 1: every pair on the float32 matrix-core instruction; 0: vector
 *                 ALU), "real_sample_half" (1: the sampled cut's scores in the filter's 16-bit arithmetic -- they only place the cut; 0: exact float32 chains), "real_second_sample" (1: a second, counting sample four times as large tightens that cut), "real_sort_lds" (1: ranked by the LDS-resident kernel when the records fit), "real_groups" (1: lists beyond the LDS ordered group by group), "real_map_lists" (0; 1: hg_map_real also writes the ranked idx / score lists), "real_whole_rounds" (3: without a cut -- R = N -- the database is cut so that k_real_select_mx's blocks fill whole rounds of that many per CU; 0: the plain geometry)
 *   ("probe_select" exists only in the measurement build, python -m hashgan_amd.build --probes) */
int hg_set_option(hg_ctx* ctx, const char* key, int64_t value);
/* Counters and facts about the last call (22 keys; the process-wide "cache_*" and "host_*" keys are listed at hg_release_cache): "optimistic_runs", "optimistic_fallbacks" (all queries rerun exactly),
 * "optimistic_requeried" (single queries rerun exactly after losing their bet), "optimistic_rebets" (second and widened bets), "last_optimistic",
 * "rank_leftovers" (queries the LDS-resident rank kernel left to the general one), "select_variant" (1 k_select, 2 k_select_dense, 3 k_select_mx,
 * 5 k_select_mx3, 6 k_select_mx4), "rank_variant" (1 k_rank_fused, 3 k_rank_cnt, 6 k_rank_lean, 7 k_rank_dense, 8 k_rank_dense<slices>), "ap_fused",
 * "cap_boost", "crowding_x100", "segments", "records_kept" (records the last bet's select left in the slices: a download, not part of a step),
 * "device_bytes", "graph_replays", "map_async_steps" / "map_async_redone" (hg_map_begin: steps enqueued blind / of those, run again by hg_map_end);
 * "cut_beyond_planes" (1: the last hg_guess_finish met a query whose cut lies beyond the b/2 + 2 planes the owner-routed exchange carries -- its bet
 * cannot be won by wider slices; a download); real-valued path: "real_attempts", "real_requeried" (queries that lost the first cut and were ranked again on their own, cumulative), "real_cap_boost", "real_path" (bit 0 filter + rescoring, bit 1 ranked in LDS,
 * bit 2 lists ordered group by group). */
int hg_get_stat(hg_ctx* ctx, const char* key, int64_t* value);
/* What hg_set_database_f32 (queries = 0) / hg_set_queries_f32 (queries = 1) found in the float table: out[0] entries outside
 * {-1, 0, +1}, out[1] zeros, out[2] minus ones, out[3] = 1 if the float table is resident on the device -- from which the caller
 * tells +-1 codes (ranked by Hamming distance), {0,1} bits and real-valued features (ranked by inner product, metric.py:13) apart. */
int hg_get_census(hg_ctx* ctx, int queries, int64_t out[4]);
/* Work buffers only grow; hg_trim frees everything except the resident code/label/feature tables
 * (stat "device_bytes" reports what the context holds) and empties the process-wide block cache (hg_release_cache): the memory
 * goes back to the HIP runtime, e.g. for another framework in the same process. */
int hg_trim(hg_ctx* ctx);
/* Device blocks, pinned host blocks and streams that a destroyed or trimmed context gives up go to a process-wide cache and the
 * next context takes them from there instead of the HIP runtime (hipMalloc now and then stalls for SECONDS on this stack, a
 * stream costs 1.5 - 18 ms to create: a caller that builds a context per evaluation -- main.py:164 builds a MAPs object per
 * evaluation -- would pay that per call).  At most HG_CACHE_MB (49152: a sixth of the part's 288 GB) of device and HG_PIN_CACHE_MB (512) of pinned memory are
 * held; hg_release_cache returns all of it to the runtime.  Stats (any context): "cache_device_bytes", "cache_pinned_bytes",
 * "cache_hits", "cache_misses", "cache_streams"; "host_us_<phase>" / "host_n_<phase>" / "host_max_us_<phase>" with phase one of
 * init, devmalloc, devfree, hostmalloc, hostfree, destroy, stream, event, sync (waiting for the GPU), pack (host packing pass), thread
 * (starting the float table's staging thread): what the process has spent there.
 * The reference has no counterpart (lib/metric.py allocates NumPy arrays per call). */
int hg_release_cache(void);
/* Pay now what a context otherwise pays on first use of each path: the second stream and the pinned staging of the float
 * tables' uploads (hg_set_database_f32), the result block, the code objects of every translation unit (the runtime loads one
 * when a kernel of it is first launched), the host packing threads.  ~30 ms; hashgan_amd's MAPs pool calls it once per context
 * it creates, so that the first evaluation of ANOTHER shape or path in a process costs what the later ones do. */
int hg_preload(hg_ctx* ctx);
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
