/*
 * shannon_hip.h -- C ABI of libshannon_hip.so: the MI355X (gfx950) implementation of the
 * Shannon RNA-Seq assembler hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference (sreeramkannan/Shannon, pure Python 2 + external programs) has no FFI of its
 * own; each entry point below replaces the reference call site / external program cited next
 * to it and is what a ctypes binding of that call site would bind.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on failure; shn_last_error()
 *     returns the message of the last failure on the calling thread.  Nothing ever blocks on
 *     stdin or calls exit() (the reference's pdb.set_trace()/raw_input() paths become errors).
 *   - inputs are borrowed for the duration of the call; outputs are caller-allocated arrays or
 *     opaque handles freed with the matching *_destroy.  Device buffers never escape except
 *     through the explicit *_device_ptr accessors (for RCCL / torch interop).
 *   - one shn_ctx per (device, stream); a context is used by one host thread at a time.
 *   - base codes: A=0 C=1 G=2 T=3; k-mers are packed big-endian 2-bit (first base most
 *     significant) in the low 2*k bits of a uint64, so integer order == string order.
 */
#ifndef SHANNON_HIP_H
#define SHANNON_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct shn_ctx shn_ctx;
typedef struct shn_reads shn_reads;
typedef struct shn_table shn_table;

#define SHN_OK 0
#define SHN_ERR_ARG -1
#define SHN_ERR_HIP -2
#define SHN_ERR_NOMEM -3
#define SHN_ERR_OVERFLOW -4
#define SHN_ERR_INTERNAL -5

#define SHN_ENC_ASCII 0 /* 'A','C','G','T' (either case); anything else is "non-ACGT" */
#define SHN_ENC_CODES 1 /* 0..3; anything else is "non-ACGT"                          */

const char* shn_last_error(void);
const char* shn_version(void);

/* ---- context ----------------------------------------------------------------------------- */
/* `stream` is a hipStream_t (NULL = the device's default stream). */
int shn_ctx_create(int device, void* stream, shn_ctx** out);
void shn_ctx_destroy(shn_ctx* ctx);
int shn_ctx_sync(shn_ctx* ctx);
/* A second context on the device of `parent` with a stream of its own (destroyed with it), for a host thread that works beside the
 * owner of `parent`; event timing is off on it.                                                                                 */
int shn_ctx_fork(const shn_ctx* parent, shn_ctx** out);
/* The device workspaces of the context's top-level stages (counting, extension, contig stage, probe table, routing, unitigs) in a
 * set of its own: by default all contexts of a process share one set, which serves one pipeline at a time -- the reference runs its
 * stages one after the other in one process (shannon.py:427-604).  A second pipeline whose stages run BESIDE the first one's
 * (two batches in flight) gives its context a set of its own first.  Forked contexts use their parent's set.                   */
int shn_ctx_own_workspaces(shn_ctx* ctx);

/* HIP-event timing on the context's stream (bench.py: per-kernel-group durations).
 * shn_timer_begin/end bracket a region under `slot` (0..31); shn_timer_ms() synchronises and
 * returns the accumulated milliseconds and number of regions since the last reset. */
int shn_timer_reset(shn_ctx* ctx);
int shn_timer_ms(shn_ctx* ctx, int slot, double* ms, uint64_t* n_regions);
/* the ALGORITHMIC bytes of the launches timed under `slot` since the last reset -- what those kernels have to read and write at the
 * least, by the byte model written next to each launch (0: the slot's launch sites declare none and bench.py prices them itself) */
int shn_timer_bytes(shn_ctx* ctx, int slot, uint64_t* bytes);
const char* shn_timer_name(int slot);

/* ---- reads: upload + 2-bit pack on device --------------------------------------------------
 * Replaces the read files the reference re-parses three times (jellyfish, shannon.py:439;
 * kmers_for_component.py:329-403; multibridging.py:22-98) and rc_gnu.py/rc_s.py strand
 * doubling (shannon.py:394-424): reverse complements are computed on chip, never stored.
 *   bytes    host pointer, concatenated read bases (encoding `enc`)
 *   offsets  host pointer to n_reads+1 byte offsets, or NULL for fixed-length reads
 *   fixed_len read length when offsets == NULL                                            */
int shn_reads_create(shn_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_reads,
                     uint32_t fixed_len, int enc, shn_reads** out);
void shn_reads_destroy(shn_reads* r);
uint64_t shn_reads_count(const shn_reads* r);
uint64_t shn_reads_total_bases(const shn_reads* r);
uint32_t shn_reads_max_len(const shn_reads* r);
/* number of reads containing a non-ACGT character */
uint64_t shn_reads_n_invalid(const shn_reads* r);

/* Host utility: dst row i = src row idx[i] (rows of row_bytes bytes; src has n_src_rows rows), on `threads` host threads.
 * Used for the capped read sets of a partition (kmers_for_component.py:322-403 writes them to per-component files;
 * here they are gathered from the resident read matrix).  SHN_ERR_ARG on an index out of range.                          */
/* Host utility: all k-windows of n_strings ACGT strings (text + offsets), in order, as packed keys (keys_out, may be NULL;
 * k <= 32) and/or k-byte rows (rows_out, may be NULL).  What k1mers2component / the per-component k1-mer files
 * (kmers_for_component.py:244-305, 452-477) enumerate.  SHN_ERR_ARG on a base outside ACGT.                              */
int shn_string_windows(const uint8_t* text, const uint64_t* off, uint64_t n_strings, int k, uint64_t* keys_out, uint8_t* rows_out);

int shn_gather_rows(const uint8_t* src, uint64_t n_src_rows, uint64_t row_bytes, const int64_t* idx, uint64_t n, uint8_t* dst, int threads);
/* Host utility: segment i of dst (dst_off[i] .. dst_off[i+1]) = segment order[i] of src (src_off has n_src + 1 entries), on
 * `threads` host threads.  Merges the candidate contigs of the ranks' shards into the global seed order (weight descending,
 * seed k1-mer ascending: the order of the loop of extension_correction.py:334-354).  SHN_ERR_ARG on an index out of range or
 * on segments of different lengths.                                                                                          */
int shn_gather_segments(const uint8_t* src, const uint64_t* src_off, uint64_t n_src, const int64_t* order, uint64_t n, uint8_t* dst,
                        const uint64_t* dst_off, int threads);
/* Host utility: a two-line FASTA from byte segments -- record i = ">" prefix i "\n" segment order[i] "\n" (the file
 * reconstructed_single_contigs.fasta of extension_correction.py:506-513 from the candidate buffer of the contig stage, without a
 * Python string per contig).  dst == NULL: only *dst_len is set.                                                               */
int shn_fasta_records(const uint8_t* src, const uint64_t* src_off, uint64_t n_src, const int64_t* order, uint64_t n, const char* prefix,
                      uint8_t* dst, uint64_t dst_cap, uint64_t* dst_len);

/* ---- (K+1)-mer counting ----------------------------------------------------------------------
 * Replaces `jellyfish count -m k1 ... ; jellyfish dump -c -t -L lower` (shannon.py:439-441;
 * duplicate call site run_MB_SF_fn.py:177-184).  Counts every ACGT-only k1-window of every
 * read of every set.  both_strands=1 folds a window and its reverse complement onto the
 * canonical (smaller) key -- the count of x in the reference's strand-doubled input equals
 * table[canon(x)] (x2 if x is its own reverse complement); both_strands=0 counts forward
 * windows only (the reference's -s / --strand_specific mode).  2 <= k1 <= 32.             */
int shn_count_k1mers(shn_ctx* ctx, shn_reads* const* sets, int n_sets, int k1, int both_strands,
                     shn_table** out);
void shn_table_destroy(shn_table* t);
uint64_t shn_table_size(const shn_table* t);          /* number of distinct stored keys          */
uint64_t shn_table_total(const shn_table* t);         /* number of windows counted               */
int shn_table_k(const shn_table* t);
int shn_table_canonical(const shn_table* t);
/* copy the table to host arrays of shn_table_size() entries (order: bucket, then key) */
int shn_table_download(shn_ctx* ctx, const shn_table* t, uint64_t* keys, uint32_t* counts);
/* expanded to the reference's k1mer.dict_org content: every key of the strand-doubled input
 * with count >= lower, sorted ascending.  Call with keys==NULL to get the size in *n.       */
int shn_table_dump(shn_ctx* ctx, const shn_table* t, uint32_t lower, uint64_t* keys, uint32_t* counts,
                   uint64_t* n);
/* `jellyfish dump -L lower` (--kmer_hard_cutoff, shannon.py:237-241, 441): a new table of the k1-mers whose count in the
 * reference's input is >= lower -- for a canonical table the count of the strand-doubled input, so a k1-mer that is its own
 * reverse complement stands for twice its stored count (the rule of shn_table_dump).  Same layout and bucket structure, built by
 * a stable compaction on the device; shn_table_total() stays the number of windows counted.  Caller destroys both tables.       */
int shn_table_filter_lower(shn_ctx* ctx, const shn_table* t, uint32_t lower, shn_table** out);
/* device-resident view for collectives (RCCL all-to-all of (key,count) shards) */
int shn_table_device_ptrs(const shn_table* t, void** keys, void** counts);
/* look up `n` keys (host array, already canonical if the table is) -> counts (0 if absent) */
int shn_table_lookup(shn_ctx* ctx, const shn_table* t, const uint64_t* keys, uint64_t n, uint32_t* counts);

/* Build a table from already-aggregated (key,count) pairs resident on the device, summing
 * duplicates (the local reduce-by-key after the all-to-all bucket exchange). */
int shn_table_from_pairs(shn_ctx* ctx, const void* dev_keys, const void* dev_counts, uint64_t n, int k1,
                         int canonical, shn_table** out);
/* The table of a strand-specific paired run (-s, shannon.py:407-411 and :436-439 without -C): the k1-mers of `fwd` (forward counting
 * of reads_1) and the reverse complements of those of `other` (forward counting of reads_2, i.e. the k1-mers of RC(reads_2)), equal
 * keys summed; both inputs plain (non-canonical) tables of one k.  On the device.                                                 */
int shn_table_merge_rc(shn_ctx* ctx, const shn_table* fwd, const shn_table* other, shn_table** out);
/* Owner rank of each stored key for hash-sharding across GPUs: fills per-rank counts
 * (host, n_ranks entries) and writes keys/counts grouped by rank into the device buffers. */
int shn_table_shard(shn_ctx* ctx, const shn_table* t, int n_ranks, uint64_t* per_rank, void* dev_keys_out,
                    void* dev_counts_out);
/* mode 0: as above; mode 1: owner = rank of the key's minimizer (m = 13; the hash of the minimizer's order value mod n_ranks) --
 * a k1-mer and its eight neighbours share their minimizer six times out of seven, so most edges of the k1-mer graph stay inside a
 * shard: the shards the components are labelled on (shn_cc_*).                                                                */
int shn_table_shard_mode(shn_ctx* ctx, const shn_table* t, int n_ranks, int mode, uint64_t* per_rank, void* dev_keys_out,
                         void* dev_counts_out);

/* ---- contig extension / k1-mer error correction ----------------------------------------------
 * Replaces the hot loop of extension_correction.run_correction (extension_correction.py:334-354:
 * load_kmers + lowComplexity filter :142-149,202-221; heaviest-first greedy extend :223-245 with
 * ties A,G,C,T :10,159-166 and a global `traversed` set).  Seeds are all k1-mers of the
 * strand-doubled input with weight >= min_weight, in (weight desc, k1-mer asc) order; walk r is
 * the contig seeded by the r-th seed, or void if an earlier walk traversed its seed.  The
 * accept filter / duplicate_check / contig graph (:358-513) run on the host over the emitted
 * contigs (shannon_amd/extension_correction.py).                                              */
typedef struct shn_ext shn_ext;
int shn_extend(shn_ctx* ctx, const shn_table* t, uint32_t min_weight, int max_iterations, shn_ext** out);
void shn_ext_destroy(shn_ext* e);
uint64_t shn_ext_n_walks(const shn_ext* e);
int shn_ext_iterations(const shn_ext* e);
uint64_t shn_ext_total_steps(const shn_ext* e);   /* walk steps executed over all fixpoint iterations */
uint64_t shn_ext_wave_steps(const shn_ext* e);    /* ... of which by the wavefront (long-walk) kernel */
uint64_t shn_ext_fresh_steps(const shn_ext* e);   /* ... of which by the thread walker in the first round of a rank block (the bulk launches) */
int shn_ext_dense_rounds(const shn_ext* e);       /* rounds whose begin / mark passes streamed every claim (the others followed line flags) */
uint64_t shn_ext_settled_walks(const shn_ext* e); /* seeds never launched: a lower-ranked k1-mer on a chain of forced links takes them first
                                                    * (extension_correction.py:223-245, 346: such a seed is always `in traversed`) */
/* Diagnostics (no counterpart in the reference, whose loop extension_correction.py:334-354 is sequential and deterministic): with
 * SHN_EXT_DIGEST=1 in the environment shn_extend keeps checksums of the arrays of its stages -- out[stage * 64 + chunk], 8 stages
 * (table keys, counts, bucket offsets, weights + flags, adjacency records, seed order, converged claims, walk records) x 64 chunks
 * of each array.  Two runs on the same input must agree everywhere; the stress test names the first stage and chunk that do not. */
int shn_ext_digests(const shn_ext* e, uint64_t* out /* [512] */);
/* process-wide debug counters: 0 blocks of the caching allocator freed twice, 1 frees of pointers it does not own, 2 (spare)       */
uint64_t shn_debug_counter(int which);
/* test hooks of the caching allocator (tests/test_allocator_gpu.py): a block on a context's stream, a fill kernel that can be made
 * slow, a read-back.  Not part of the path; there so that the stream ordering of freed blocks can be tested from outside.        */
int shn_debug_alloc(shn_ctx* ctx, uint64_t bytes, void** out);
void shn_debug_free(shn_ctx* ctx, void* p);
int shn_debug_fill(shn_ctx* ctx, void* p, uint64_t n_words, uint32_t value, uint32_t spin);
int shn_debug_read(shn_ctx* ctx, const void* p, uint64_t n_words, uint32_t* host_out);
/* per walk (host arrays of shn_ext_n_walks entries): right/left extension lengths (n_right ==
 * 0xFFFFFFFF marks a void walk) and the weight sum including the seed (tot_wt, :351)          */
int shn_ext_stats(shn_ctx* ctx, const shn_ext* e, uint32_t* n_right, uint32_t* n_left, uint64_t* tot_weight);
/* The same for the walks lo <= rank < lo + n only.                                                                          */
int shn_ext_stats_range(shn_ctx* ctx, const shn_ext* e, uint64_t lo, uint64_t n, uint32_t* n_right, uint32_t* n_left, uint64_t* tot_weight);
/* Pipelining hook (per calling thread; NULL clears it): shn_extend / shn_extend_sharded call `cb(user, ext, lo, hi, status)` on
 * the calling thread whenever the walks lo <= rank < hi are final -- the reference's loop (extension_correction.py:334-369)
 * finishes contigs in exactly this order, so the caller can run duplicate_check / contig_connections (shn_cgraph_add) on them
 * while the later, lighter seeds are still being walked.  status: 0 a block, 1 the last block, -1 everything handed over so
 * far is void (the fixpoint audit reopened the blocks).  `ext` is valid for shn_ext_stats_range / shn_ext_emit /
 * shn_ext_seed_info during the call-back only.                                                                                */
typedef void (*shn_block_cb)(void* user, shn_ext* ext, uint64_t lo, uint64_t hi, int status);
void shn_ext_set_block_callback(shn_block_cb cb, void* user);
/* contig strings (ASCII) of the selected walk ranks, contig i at bases_out[offsets[i]..offsets[i+1]) */
int shn_ext_emit(shn_ctx* ctx, const shn_ext* e, const uint32_t* ranks, uint64_t n_sel, const uint64_t* offsets,
                 uint8_t* bases_out);
/* weight (count in the strand-doubled input, 0 if absent or low-complexity) of k1-mer strings */
int shn_ext_weights(shn_ctx* ctx, const shn_ext* e, const uint64_t* keys, uint64_t n, uint32_t* weights);
/* The non-void walks only, in seed order (rank ascending): rank[], n_right[], n_left[], tot_weight[] -- the inputs of the
 * accept filter of extension_correction.py:361.  Call with rank == NULL to get *n_live, then with arrays of that size
 * (*n_live = capacity on entry).                                                                                       */
/* Sharded variant: the connected components of the k1-mer graph are dealt to `world` ranks (the large ones balanced by
 * size, the rest by hash; every rank computes the same assignment from the same table) and only the seeds of `rank`'s
 * components walk.  Walks never leave their component, so the union of the shards' walks is the unsharded result.
 * shn_ext_seed_info: seed k1-mer (packed) and seed weight of walks -- (weight desc, k1-mer asc) is the global walk order
 * that merges the shards (extension_correction.py:334-345).                                                             */
int shn_extend_sharded(shn_ctx* ctx, const shn_table* t, uint32_t min_weight, int max_iterations, int world, int rank, shn_ext** out);
int shn_ext_seed_info(shn_ctx* ctx, const shn_ext* e, const uint32_t* ranks, uint64_t n, uint64_t* keys, uint32_t* weights);
/* ---- component labelling on owner shards: the N-rank form of the reference's one shared k1-mer dictionary
 * (extension_correction.py:202-221 loads ALL k1-mers into one dict; shannon.py:527-566 fans out only after that).  No rank ever
 * holds the whole table: every rank labels the shard of k1-mers it owns (shn_table_shard_mode 1), asks the owners of the
 * neighbours it cannot see, and the component graph -- small: one node per local component with an edge to another rank -- is
 * solved on every rank from the gathered edges.  Then whole components travel to their owner rank, which runs the unsharded
 * shn_extend on a table of its own.  All buffers are device memory of the caller; every call returns with its work done.
 *   shn_cc_create        local components of the shard `t` (which must outlive the object); counts the queries
 *   shn_cc_query_counts  queries per destination rank (only higher ranks are asked: an edge is seen from both ends)
 *   shn_cc_queries       the queries grouped by destination: neighbour / sibling key (8 bytes), the asker's local root (4 bytes)
 *   shn_cc_answer        the queries received (grouped by source rank, recv_per_rank[s] each) against this shard: edges as pairs of
 *                        global ids = base[rank] + local root (base = exclusive sums of the ranks' shard sizes); room for 2 ids
 *                        per query
 *   shn_cc_solve         n_edges pairs of ids (all ranks' edges) -> the distinct ids ascending + for each the smallest id of its
 *                        component (room for 2 n_edges each); independent of the order of the edges; ids < id_limit (0: unknown)
 *   shn_cc_labels        global label of every k1-mer of the shard, table order (8 bytes each)
 *   shn_cc_owners        owner rank of every k1-mer (1 byte each): hash of its label mod world, except the n_big labels listed
 *                        (ascending) with their ranks -- the components the caller balances by size
 *   shn_cc_shard         the shard's (key, count) pairs grouped by owner rank (per_rank[r] each)                               */
/* diagnostics of the labelling kernel (SHN_CC_DEBUG=1): out4[0] = look-ups that found their key since the last reset                */
int shn_debug_cc_counters(uint64_t* out4, int reset);
typedef struct shn_cc shn_cc;
int shn_cc_create(shn_ctx* ctx, const shn_table* t, int world, int rank, shn_cc** out);
void shn_cc_destroy(shn_cc* c);
int shn_cc_query_counts(const shn_cc* c, uint64_t* per_rank);
int shn_cc_queries(shn_cc* c, void* dev_keys_out, void* dev_labs_out);
int shn_cc_answer(shn_cc* c, const void* dev_keys, const void* dev_labs, const uint64_t* recv_per_rank, const uint64_t* base,
                  void* dev_edges_out, uint64_t* n_edges);
int shn_cc_solve(shn_ctx* ctx, const void* dev_edges, uint64_t n_edges, uint64_t id_limit, void* dev_nodes_out, void* dev_labels_out,
                 uint64_t* n_nodes);
int shn_cc_labels(shn_cc* c, uint64_t base_me, const void* dev_nodes, const void* dev_labels, uint64_t n_nodes, void* dev_glabel_out);
int shn_cc_owners(shn_cc* c, const void* dev_glabel, const void* dev_big, const void* dev_big_owner, uint64_t n_big, void* dev_owner_out);
int shn_cc_shard(shn_cc* c, const void* dev_owner, uint64_t* per_rank, void* dev_keys_out, void* dev_counts_out);
int shn_ext_live_stats(shn_ctx* ctx, const shn_ext* e, uint64_t* n_live, uint32_t* rank, uint32_t* n_right, uint32_t* n_left,
                       uint64_t* tot_weight);
/* ... restricted to walks of at least min_steps steps (contig length k1 + steps): the first clause of the accept filter,
 * extension_correction.py:361, applied before the download.                                                                      */
int shn_ext_live_stats_min(shn_ctx* ctx, const shn_ext* e, uint32_t min_steps, uint64_t* n_live, uint32_t* rank, uint32_t* n_right,
                           uint32_t* n_left, uint64_t* tot_weight);
/* The accept filter of the extension loop (extension_correction.py:361: len >= min_length and len * avg_weight ** 0.25 >= threshold,
 * threshold = 2 * min_length * min_weight ** 0.25) over the non-void walks, in seed order: rank, steps (= n_right + n_left; the contig
 * has k1 + steps bases), tot_weight and cls per candidate -- cls 1: passes for sure, 2: within 1e-9 (relative) of the threshold, for
 * the caller to decide with the reference's own arithmetic.  *n_out: in = room of the arrays (NULL arrays: sizing call), out = count. */
int shn_ext_accept(shn_ctx* ctx, const shn_ext* e, uint32_t min_length, double threshold, uint64_t* n_out, uint32_t* rank, uint32_t* steps,
                   uint64_t* tot_weight, uint8_t* cls);

/* Host-side (CPU, native) contig bookkeeping of run_correction over the contigs emitted above:
 * duplicate_check (extension_correction.py:247-270, r=15, f=0.5) and the contig graph by shared
 * K-mers (:372-397), sequential in seed order as the reference.  accepted_out[i] = 1-based accepted
 * index of candidate i or 0.  Call once with conn_nb == NULL to get *n_acc_out / *n_conn, then again
 * with arrays conn_off[n_acc+1], conn_nb[n_conn], conn_w[n_conn]: neighbours of accepted contig a in
 * the reference's dict insertion order.                                                          */
int shn_contig_graph(const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                     int32_t* accepted_out, uint64_t* n_acc_out, uint64_t* conn_off, int32_t* conn_nb, int32_t* conn_w,
                     uint64_t* n_conn);
/* hit count of the `best` contig (max_till_now, :255-259; 0 = no shared r-mer) of every candidate of the calling thread's
 * last shn_contig_graph call                                                                                            */
int shn_contig_best_counts(int32_t* out, uint64_t n_cand);

/* The same stage fed in several calls (candidates still in seed order): the pipelined extension hands over the candidates
 * of every rank block as soon as that block is final.  shn_cgraph_add: accepted_out[i] = 1-based accepted index (counted
 * over all calls) or 0; best_counts_out may be NULL.  Sizes, then export of the connections as in shn_contig_graph.        */
typedef struct shn_cgraph shn_cgraph;
int shn_cgraph_create(int k1, int r, double f, shn_cgraph** out);
void shn_cgraph_destroy(shn_cgraph* g);
int shn_cgraph_add(shn_cgraph* g, const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int32_t* accepted_out, int32_t* best_counts_out);
int shn_cgraph_sizes(const shn_cgraph* g, uint64_t* n_acc, uint64_t* n_conn);
int shn_cgraph_export(const shn_cgraph* g, uint64_t* conn_off, int32_t* conn_nb, int32_t* conn_w);

/* The same stage in one call with the bulk of the work on the GPU (csrc/contig_gpu.hip), for inputs with hundreds of thousands of
 * candidates: the r-mers of all candidates are sorted on the device and candidates sharing an r-mer are clustered -- duplicate_check
 * (extension_correction.py:247-270) only ever compares a candidate with accepted contigs of its own cluster, so the clusters are
 * decided independently (in seed order inside a cluster); contig_connections (:372-397) come from a device sort of the accepted
 * contigs' K-mers, replayed on the host over the K-mers that occur in two contigs.  Same outputs as shn_cgraph_add on a fresh
 * graph; *out is read with shn_cgraph_sizes / shn_cgraph_export and freed with shn_cgraph_destroy.                              */
int shn_contig_stage(shn_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                     int32_t* accepted_out, int32_t* best_counts_out, shn_cgraph** out);
/* The same two steps without the candidates' text crossing the bus twice: shn_ext_emit_device leaves the text of the selected walks
 * (extension_correction.py:334-354, the contigs before the accept decisions of :358-397) on the device as a shn_devtext,
 * shn_contig_stage_device reads it there, shn_devtext_segments fetches the pieces idx[0..n_idx) of the n_off - 1 pieces the offsets
 * cut it into (the accepted contigs), one after the other.                                                                     */
typedef struct shn_devtext shn_devtext;
int shn_ext_emit_device(shn_ctx* ctx, const shn_ext* e, const uint32_t* ranks, uint64_t n_sel, const uint64_t* offsets, shn_devtext** out);
int shn_contig_stage_device(shn_ctx* ctx, const shn_devtext* text, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                            int32_t* accepted_out, int32_t* best_counts_out, shn_cgraph** out);
int shn_devtext_segments(shn_ctx* ctx, const shn_devtext* text, const uint64_t* off, uint64_t n_off, const int64_t* idx, uint64_t n_idx, uint8_t* out);
void shn_devtext_destroy(shn_devtext* t);
/* Connected components of the contig graph by the reference's depth-first search (extension_correction.py:417-434) over the
 * connections CSR of shn_cgraph_export: comp_of[a-1] = root contig of contig a (1-based), members[] = the components' contigs in the
 * order the DFS pops them, component j at members[comp_off[j]..comp_off[j+1]), comp_edges[j] = its number of undirected edges
 * (the E of the METIS header, :446-456).  comp_off: n_acc+1 entries, comp_edges: n_acc entries.                               */
int shn_contig_components(uint64_t n_acc, const uint64_t* conn_off, const int32_t* conn_nb, int32_t* comp_of, int32_t* members,
                          uint64_t* comp_off, uint64_t* comp_edges, uint64_t* n_comp_out);

/* ---- read -> partition routing -----------------------------------------------------------------
 * Replaces the read-streaming loops of kmers_for_component (kmers_for_component.py:322-403;
 * get_rmers :186-192, get_comps :194-205).  `probe` maps every k1-mer of every partition's
 * contigs (plain strings, k1mers2component :244-305) to value = set_id+1; set s holds partition
 * ids set_members[set_off[s]..set_off[s+1]).  Doubled read index d: SE d<N -> R[d], d>=N ->
 * RC(R[d-N]); PE d<N -> (R1[d], RC(R1[d])), d>=N -> (RC(R2[d-N]), R2[d-N]) (shannon.py:396-424 as
 * written).  Result: (partition id, d) pairs sorted by partition then d == the order of the
 * reference's reads{comp}.fasta files; reads with a non-ACGT base are dropped (:336,:376).   */
typedef struct shn_routes shn_routes;
int shn_table_create(shn_ctx* ctx, const uint64_t* keys, const uint32_t* values, uint64_t n, int k, int canonical,
                     shn_table** out);
/* k1mers2component on the GPU (kmers_for_component.py:244-305): the probe table + partition sets from the partition contigs in one
 * call (csrc/probe_gpu.hip).  bases / off: the contigs of all partitions one after the other (ASCII), part_of[c] = partition of
 * contig c (ascending).  Set p (p < n_parts) is {p}; the sets of k1-mers that occur in several partitions follow.              */
typedef struct shn_probe shn_probe;
int shn_probe_build(shn_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint64_t n_contigs, const uint32_t* part_of, uint32_t n_parts,
                    int k1, shn_probe** out);
void shn_probe_destroy(shn_probe* p);
const shn_table* shn_probe_table(const shn_probe* p);        /* owned by the probe */
uint32_t shn_probe_n_sets(const shn_probe* p);
uint64_t shn_probe_n_members(const shn_probe* p);
int shn_probe_sets(const shn_probe* p, uint32_t* set_off /* n_sets + 1 */, uint32_t* set_mem /* n_members */);
int shn_route_reads(shn_ctx* ctx, const shn_reads* r1, const shn_reads* r2, int k1, const shn_table* probe,
                    const uint32_t* set_off, const uint32_t* set_members, uint32_t n_sets, shn_routes** out);
/* The same for a strand-specific run (-s / --ss / --strand_specific, shannon.py:407-411): the read files the routing loops stream
 * (kmers_for_component.py:322-403, always with double_stranded = False after shannon.py:427) are then `reads` (single-end) or
 * reads_1 and RC(reads_2) (paired), NOT strand-doubled -- strand_specific != 0: a route (partition, d) has d < N, a pair is
 * (r1[d], RC(r2[d])) and matches through the k1-mers of r1[d] and of RC(r2[d]).  strand_specific == 0: shn_route_reads.          */
int shn_route_reads_mode(shn_ctx* ctx, const shn_reads* r1, const shn_reads* r2, int k1, const shn_table* probe,
                         const uint32_t* set_off, const uint32_t* set_members, uint32_t n_sets, int strand_specific, shn_routes** out);
void shn_routes_destroy(shn_routes* r);
uint64_t shn_routes_size(const shn_routes* r);
int shn_routes_download(shn_ctx* ctx, const shn_routes* r, uint32_t* pid, uint32_t* ridx);
/* Routes are sorted by (partition, doubled read index).  start[p] (n_parts+1 entries) = first route of partition p;
 * below[p] = how many routes of partition p have a read index < split (the forward half of the strand-doubled order).
 * With shn_routes_download_range a caller fetches only the capped prefix a partition's graph may consume
 * (multibridging.py:26-30) instead of every route.                                                                       */
int shn_routes_bounds(shn_ctx* ctx, const shn_routes* r, uint32_t n_parts, uint32_t split, uint64_t* start, uint64_t* below);
int shn_routes_download_range(shn_ctx* ctx, const shn_routes* r, uint64_t lo, uint64_t n, uint32_t* ridx);

/* ---- K-mer seed scans of reads against graph nodes -----------------------------------------------
 * Replace the per-read Python loops of Read.find_bridging_reads (mbgraph.py:88-111) and known_paths
 * (mbgraph.py:1355-1388).  `patterns`: plain K-mer keys -> value = id+1 (shn_table_create).
 * shn_seed_scan: every read, every start in [1, len-K) (the reference's range(1, len(read)-K)) whose
 * K-mer is a key -> (read, start, id), in (read, start) order; call with out_read == NULL for *n_hits.
 * shn_seed_ends: table value (0 = absent) of the first and of the last K-mer of every read.      */
int shn_seed_scan(shn_ctx* ctx, const shn_reads* reads, int K, const shn_table* patterns, uint64_t* n_hits,
                  uint32_t* out_read, uint32_t* out_start, uint32_t* out_id);
/* r-mer join of two sets of sequences on the device: every window of every sequence of `cands` against every occurrence
 * of the same r-mer in the sequences of `foreign` -> (candidate, window start, foreign sequence) triples in (candidate,
 * start) order.  Call with out_cand == NULL for *n_hits.  Guard of the component-sharded duplicate_check. */
int shn_rmer_join(shn_ctx* ctx, const shn_reads* cands, const shn_reads* foreign, int r, uint64_t* n_hits, uint32_t* out_cand,
                  uint32_t* out_start, uint32_t* out_foreign);
int shn_seed_ends(shn_ctx* ctx, const shn_reads* reads, int K, const shn_table* patterns, uint32_t* first_id,
                  uint32_t* last_id);

/* ---- multibridged de-Bruijn graph of one partition (host, native) ----------------------------------
 * Replaces multibridging.main for one partition (multibridging.py:327-400: load_single_jellyfish
 * :145-172, load_reads / load_mated_reads :22-97, run :209-269, output_components :271-325 over the
 * live methods of mbgraph.py).  `rows`: the n_rows k1-mers (K+1 bytes each, no separator) of the
 * partition's k1mer file in file order; reads: ASCII, r_off[n_reads+1] (the first 10*#K-mer-nodes+1 are
 * used, multibridging.py:26-30); Read.L = length of the first read.  The result is the content of
 * single_nodes.txt and nodes/edges/paths{c}.txt, flattened: see shn_graph_sizes / shn_graph_export and
 * shannon_amd/mbgraph_native.py.                                                                  */
typedef struct shn_graph shn_graph;
int shn_mbgraph_run(shn_ctx* ctx /* NULL: K-mer seed scans on the host instead of the GPU */, int K, const uint8_t* rows,
                    uint64_t n_rows, const uint8_t* r1, const uint64_t* r1_off, const uint8_t* r2, const uint64_t* r2_off,
                    uint64_t n_reads, int paired, int enc /* SHN_ENC_* */, const uint8_t* rc1, const uint8_t* rc2 /* optional:
                    1 = use the reverse complement of read i (strand-doubled view of the input) */, shn_graph** out);
void shn_graph_destroy(shn_graph* g);
/* Unitig contraction of the raw K-mer graphs of ALL partitions in one batch on the GPU (csrc/graph_gpu.hip): load_single_jellyfish
 * (multibridging.py:145-172) + the first Node.condense_all (mbgraph.py:479-498, Edge.condense :184-257).  bases / off: the contigs of
 * all partitions one after the other (ASCII; a partition's contigs in the order of its k1-mer file), part_of[c] = partition of contig c
 * (ascending).  The nodes come out in the creation order of the sequential code and every node's edge lists in the order its merges
 * leave them, so the graph stage continues bit-identically (shn_mbgraph_run_unitigs; SHN_GRAPH_CHECK=1 builds both and compares).    */
typedef struct shn_unitigs shn_unitigs;
int shn_unitigs_build(shn_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint64_t n_contigs, const uint32_t* part_of, uint32_t n_parts,
                      int K, shn_unitigs** out);
void shn_unitigs_destroy(shn_unitigs* u);
/* distinct K-mers of partition `part` (the node count after loading: the read cap is 10 x this, multibridging.py:26-30, 385-391) */
uint64_t shn_unitigs_n_kmers(const shn_unitigs* u, uint32_t part);
/* shn_mbgraph_run on partition `part` of `ug`.  rows / n_rows (may be NULL / 0): the partition's k1-mers, needed only for a partition
 * that holds a cycle of condensable edges (built by the sequential code) and for SHN_GRAPH_CHECK=1.                               */
int shn_mbgraph_run_unitigs(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const uint8_t* r1,
                            const uint64_t* r1_off, const uint8_t* r2, const uint64_t* r2_off, uint64_t n_reads, int paired, int enc,
                            const uint8_t* rc1, const uint8_t* rc2, shn_graph** out);
/* sizes[9] = n_singles, single bases, n_components, n_nodes, node bases, n_edges, n_paths, path ids, info ints */
int shn_graph_sizes(const shn_graph* g, uint64_t* sizes);
int shn_graph_export(const shn_graph* g, uint64_t* s_off, uint8_t* s_bases, double* s_cc, double* s_norm,
                     uint64_t* comp_node_off, uint64_t* comp_edge_off, uint64_t* comp_path_off, uint64_t* n_off,
                     uint8_t* n_bases, double* n_cc, uint8_t* n_cc_int, double* n_norm, int32_t* e_in, int32_t* e_out,
                     int32_t* e_w, double* e_cc, double* e_norm, uint64_t* p_off, int32_t* p_ids, int32_t* info);

/* shn_mbgraph_run_unitigs with the partition's reads known as rows of the resident input: src_a / src_b = the packed read sets of the
 * run (src_b NULL for single-end), didx[i] = doubled read index of read i (the numbering of shn_route_reads).  The device copy of
 * the distinct reads the seed scans need is then gathered from the resident sets (shn_reads_gather) instead of uploaded as text.  */
int shn_mbgraph_run_resident(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const shn_reads* src_a,
                             const shn_reads* src_b, const uint32_t* didx, const uint8_t* r1, const uint64_t* r1_off, const uint8_t* r2,
                             const uint64_t* r2_off, uint64_t n_reads, int paired, int enc, const uint8_t* rc1, const uint8_t* rc2,
                             shn_graph** out);
/* The same with the reads named by their rows only: host_a / host_b = the run's reads as host code matrices (uint8 codes 0..3,
 * [reads of the set][read length]; what src_a / src_b hold packed), didx / n_reads as above.  load_reads / load_mated_reads
 * (multibridging.py:185-236) then run as shn_reads_dedup on the device; the host decodes the text of the distinct reads from
 * the matrices.  Fails on a routed read with a base outside ACGT (routed reads have none, kmers_for_component.py:336,376).     */
int shn_mbgraph_run_rows(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const shn_reads* src_a,
                         const shn_reads* src_b, const uint8_t* host_a, const uint8_t* host_b, const uint32_t* didx, uint64_t n_reads,
                         int paired, shn_graph** out);
/* shn_mbgraph_run_rows with the routed reads named once more by where they already lie on the device: entries [route_lo, route_lo +
 * n_reads) of `routes` (shn_route_reads' result: the reads{comp}.fasta lists of kmers_for_component.py:322-403, never written); didx
 * = the same indices on the host.  The duplicate search (multibridging.py:185-236) then reads the list in place.                    */
int shn_mbgraph_run_routes(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const shn_reads* src_a,
                           const shn_reads* src_b, const uint8_t* host_a, const uint8_t* host_b, const uint32_t* didx, const shn_routes* routes,
                           uint64_t route_lo, uint64_t n_reads, int paired, shn_graph** out);
/* The distinct reads among the slots of a partition's routed reads (slot j = read j / nm of didx, mate j % nm; nm = 2 if paired),
 * numbered in order of first occurrence as Read.reads numbers them (mbgraph.py:56-70): slot_out[id] = first slot, count_out[id] =
 * copies, and for pairs role_out[id] (1 / 2) and mate_out[id] of the read's LAST occurrence (multibridging.py:214-236; single-end:
 * 0 / -1).  Host output arrays need room for n * nm entries.                                                                     */
int shn_reads_dedup(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* didx, uint64_t n, int paired, uint64_t* n_distinct,
                    uint32_t* slot_out, uint32_t* count_out, int32_t* mate_out, uint8_t* role_out);
/* A read file's text straight into a packed read set (the reference reads every second line of its 2-line FASTA
 * in Python, rc_gnu.py:15-20, kmers_for_component.py:329-403): text / n_bytes = the file (2-line FASTA records or 4-line FASTQ
 * records; format 0 = by the first character, 1 FASTA, 2 FASTQ).  Host threads parse byte ranges into pinned double buffers
 * while the previous group is copied and packed.  *n_reads_out / *read_len_out are always set; codes_out (optional, codes_cap
 * bytes) receives the [n_reads][read_len] code matrix (0..3, 4 = other); out NULL = scan / matrix only.  Reads of different
 * lengths, multi-line FASTA and malformed records are refused (SHN_ERR_ARG, message "shn_reads_ingest: unsupported: ...").      */
int shn_reads_ingest(shn_ctx* ctx, const uint8_t* text, uint64_t n_bytes, int format, uint8_t* codes_out, uint64_t codes_cap,
                     uint64_t* n_reads_out, uint32_t* read_len_out, shn_reads** out);
/* A read file shared out by bytes (the N-rank CLI; the reference streams the file once, 10 M reads at a time:
 * kmers_for_component.py:322-403): the records that START in bytes [lo, hi) of the text -- *first_out = the offset of the first one
 * (n_bytes if none), *n_records_out their number.  Record starts are found as shn_reads_ingest's threaded scan finds them ('>' at a
 * line start; '@' at a line start with a "+" line two lines on).  index_out (optional, index_cap entries of room): the offsets of
 * every stride-th of those records (the 0th, stride-th, ...), *n_index_out of them -- a sparse index from which any record of the
 * range is at most stride records away (shn_text_skip_records).  No device call.                                                    */
int shn_text_records_in_range(const uint8_t* text, uint64_t n_bytes, uint64_t lo, uint64_t hi, int format, uint64_t* first_out,
                              uint64_t* n_records_out, uint64_t stride, uint64_t* index_out, uint64_t index_cap, uint64_t* n_index_out);
/* ... and the start of the k-th record after the record starting at byte `from` (n_bytes if the text has fewer).                   */
int shn_text_skip_records(const uint8_t* text, uint64_t n_bytes, uint64_t from, uint64_t k, int format, uint64_t* offset_out);
/* The same for reads of ANY lengths (the reference's Samples/SE_read.fasta: 48-51 bases; kmers_for_component.py:329-403 and
 * multibridging.py:22-98 take whatever the lines hold): codes_out receives the reads' codes one after the other (codes_cap bytes of
 * room; the text's size always suffices), offsets_out their n_reads + 1 offsets.  Bases outside ACGT are kept as code 4 (the packed
 * set marks them and the reads that hold them, as shn_reads_create does).  out == NULL and codes_out == NULL: scan only.           */
int shn_reads_ingest_ragged(shn_ctx* ctx, const uint8_t* text, uint64_t n_bytes, int format, uint8_t* codes_out, uint64_t codes_cap,
                            uint64_t* offsets_out, uint64_t offsets_cap, uint64_t* n_reads_out, uint32_t* max_len_out,
                            uint64_t* total_bases_out, shn_reads** out);
/* The whole final merge over the text of all_reconstructed.fasta: process_concatenated_fasta.py:6-32 (rename repeated names, drop
 * sequences of < 200 bases and sequences seen before on either strand), the length sort of shannon.py:603 and shn_find_reps
 * (faster_reps.py:60-131).  The survivors in sorted order: shn_post_count / _sizes / _export (names, name_off[n+1], seqs,
 * seq_off[n+1]).  SHN_ERR_ARG on a base outside ACGT (as shn_find_reps) or an empty line.  `ds` is the strandedness of the run
 * (0 with -s / --ss): process_concatenated_fasta.py gets it (shannon.py:596); faster_reps.py always runs with -d (:604).          */
typedef struct shn_post shn_post;
int shn_post_finalize(const uint8_t* text, uint64_t n_bytes, int ds, int r, shn_post** out);
/* ... over the concatenation of several buffers (per-partition FASTA texts) */
int shn_post_finalize_bufs(const uint8_t* const* bufs, const uint64_t* lens, uint64_t n_bufs, int ds, int r, shn_post** out);
/* The same with the two passes over the bases of the transcripts on the device of `ctx` (csrc/post_gpu.hip; SURVEY 8f row 1:
 * process_concatenated_fasta.py:26-31 and faster_reps.py:104-116 on the GPU): 128-bit fingerprints of every sequence line and of
 * its reverse complement (every match is confirmed on the text), and the scan of the surviving records for the occurrences of
 * their first / last r-mers.  Names, first-come-first-served and the containment rule stay sequential host code.  Same result
 * as shn_post_finalize_bufs, byte for byte.                                                                                    */
int shn_post_finalize_dev(shn_ctx* ctx, const uint8_t* const* bufs, const uint64_t* lens, uint64_t n_bufs, int ds, int r, shn_post** out);
/* The same merge fed piece by piece: piece `index` of all_reconstructed.fasta (shannon.py:584-595 concatenates the partitions'
 * reconstructed.fasta files and the single contigs) is handed over as soon as it exists, by any host thread -- its lines are found,
 * its bytes uploaded, its sequences fingerprinted right then; shn_post_stream_finish runs the order-dependent rules
 * (process_concatenated_fasta.py:6-32, the sort of shannon.py:603, faster_reps.py:60-131) over the pieces in index order.  A piece
 * must end its last line and stay in place until the stream is finished or destroyed; capacity = bytes of device text to reserve
 * (shn_post_stream_add returns SHN_ERR_OVERFLOW beyond it).  Result: as shn_post_finalize_dev over the pieces in index order.      */
typedef struct shn_post_stream shn_post_stream;
int shn_post_stream_begin(shn_ctx* ctx, uint64_t capacity, shn_post_stream** out);
int shn_post_stream_add(shn_post_stream* ps, uint64_t index, const uint8_t* text, uint64_t n_bytes);
int shn_post_stream_finish(shn_post_stream* ps, int ds, int r, shn_post** out);
void shn_post_stream_destroy(shn_post_stream* ps);
uint64_t shn_post_count(const shn_post* p);
int shn_post_sizes(const shn_post* p, uint64_t* name_bytes, uint64_t* seq_bytes);
int shn_post_export(const shn_post* p, uint8_t* names, uint64_t* name_off, uint8_t* seqs, uint64_t* seq_off);
void shn_post_destroy(shn_post* p);
/* known_paths' test of every read (mbgraph.py:1355-1388, 114-160) on the device: reads = the partition's distinct reads, node_bases /
 * node_off = the texts of its nodes one after the other.  state_out[r]: 0 nothing to do (first or last K-mer in no node, or no
 * occurrence matches), 1 the read lies inside node node_out[r] (index into the given order), 2 the read runs past the end of a
 * node it matches and has to be searched on the host; with offset_out != NULL also 3: the same, and the read's first K-mer occurs
 * exactly once in the nodes -- in node node_out[r] at offset offset_out[r] (the host needs no index of its own for these).
 * K <= 31, ACGT only.                                                                                                            */
int shn_known_paths_scan(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off, uint64_t n_nodes,
                         uint8_t* state_out, int32_t* node_out, uint32_t* offset_out);
/* The same, and the reads of state 3 searched on the device (search_sequence, mbgraph.py:114-160): edge_off[n_nodes + 1] / edge_dst /
 * edge_ov = the out-edges of the nodes in list order (destination = index into the node order, the offset into the destination at
 * which it continues the source).  A read searched there gets state 4; its paths are records [read, length, node indices ...] in
 * paths_out (paths_cap 32-bit words), in the order the reference's recursion enumerates them; a reader walks the records by their
 * lengths up to *paths_used words and stops at a length of 0.  Reads whose search is deeper than 12 nodes or finds no room keep
 * state 3 (the caller searches them itself).                                                                                        */
int shn_known_paths_search(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off, uint64_t n_nodes,
                           const uint32_t* edge_off, const uint32_t* edge_dst, const uint32_t* edge_ov, uint8_t* state_out, int32_t* node_out,
                           uint32_t* offset_out, int32_t* paths_out, uint64_t paths_cap, uint64_t* paths_used);
/* Host threads the library keeps busy at most: min(hardware threads, affinity mask, cgroup CPU quota) / ranks on the node
 * (LOCAL_WORLD_SIZE or SHN_LOCAL_RANKS); SHN_HOST_CPUS overrides.
 * (-- ; the reference takes its process count from --nprocs, shannon.py:99.)                                                   */
/* ---- row a8: the partitioner of contig-graph components larger than --partition.  Replaces the external call
 * `gpmetis -ufactor=U componentN.txt P` of kmers_for_component.py:221,234 (METIS 5: randomised, unpinned -- the parity cases
 * replay a given partition vector) by a deterministic multilevel k-way partitioner in the library (csrc/partition_host.hip):
 * heavy-edge matching, growth on the coarsest graph, boundary refinement per level, balance (1 + U/1000) n / P.
 * shn_partition_metis reads the reference's componentN.txt text (extension_correction.py:446-456); shn_partition_csr the same
 * graph as CSR; part_out[v] in [0, n_parts).  shn_metis_reweight = weight_updated_graph.py:24-42 (edges cut by `part` times
 * `penalty`, the reference's text format); out == NULL: only *out_len.                                                        */
int shn_partition_metis(const char* text, uint64_t len, uint64_t n_vertices, int n_parts, int ufactor, int32_t* part_out);
int shn_partition_csr(uint64_t n_vertices, const uint64_t* off, const int32_t* nb, const int64_t* w, int n_parts, int ufactor, int32_t* part_out);
int shn_metis_reweight(const char* text, uint64_t len, const int32_t* part, uint64_t n_vertices, int penalty, char* out, uint64_t out_cap, uint64_t* out_len);
int shn_host_cpus(void);
/* OPT-IN, process-wide: glibc's allocator serves blocks up to 32 MB from the heap and never trims it (mallopt), so that the host
 * stages do not pay for fresh zero pages every batch.  For programs that own their process (bench.py, shannon.py); also applied
 * when the library is loaded with SHN_MALLOC_TUNE=1 in the environment.  An embedding application is otherwise left alone.   */
void shn_malloc_tune_now(void);
/* Rows of resident fixed-length read sets as a new read set: read i = row rows[i] of set a (flags[i] bit 0 clear) or b (set),
 * reverse-complemented if bit 1 is set; the selected rows must hold ACGT only.                                                   */
int shn_reads_gather(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* rows, const uint8_t* flags, uint64_t n,
                     shn_reads** out);
/* The inverse of shn_graph_export (sizes[9] as shn_graph_sizes; arrays as shn_graph_export without `info`): a graph object from
 * flattened nodes / edges / paths tables, e.g. the reference's own files.                                                       */
int shn_graph_from_tables(const uint64_t* sizes, const uint64_t* s_off, const uint8_t* s_bases, const double* s_cc, const double* s_norm,
                          const uint64_t* comp_node_off, const uint64_t* comp_edge_off, const uint64_t* comp_path_off, const uint64_t* n_off,
                          const uint8_t* n_bases, const double* n_cc, const uint8_t* n_cc_int, const double* n_norm, const int32_t* e_in,
                          const int32_t* e_out, const int32_t* e_w, const double* e_cc, const double* e_norm, const uint64_t* p_off,
                          const int32_t* p_ids, shn_graph** out);

/* ---- sparse-flow transcript reconstruction (host, native, over the LP kernel below) ------------------------------------------
 * Replaces algorithm_SF.py for every component of every partition (algorithm_SF.py:74-613: ParseNodeFile / ParseEdgeFile /
 * ParseKnownPathsFile, findStartAndEnd2, algorithm2 with path_decompose, read_Y_paths, single_nodes_to_fasta; one process per
 * component in the reference, run_MB_SF_fn.py:239-254).  graphs[g] = the graph of partition g, snames[g] = "<sample>_<partition>".
 * All components advance together; the decompositions that need LP trials form one shn_lp_solve_batch per round (component c
 * of a partition draws its costs from the problem ids (c << 20) + call number).  Text g = reconstructed_comp_*.fasta of
 * partition g concatenated in component order, then reconstructed_comp_-1.fasta (its single nodes).                              */
typedef struct shn_sflow shn_sflow;
int shn_sparse_flow(shn_ctx* ctx, const shn_graph* const* graphs, uint32_t n_graphs, const char* const* snames, uint64_t seed, shn_sflow** out);
/* The same as one of several calls host threads make at once (run_MB_SF_fn.py:242-250 starts one algorithm_SF.py process per
 * component; here: one call per partition, right behind its graph stage): the LP batches go to the calling thread's own stream.  The
 * results equal those of ONE shn_sparse_flow over all the graphs.                                                              */
int shn_sparse_flow_thread(shn_ctx* ctx, const shn_graph* const* graphs, uint32_t n_graphs, const char* const* snames, uint64_t seed, shn_sflow** out);
void shn_sflow_destroy(shn_sflow* s);
uint64_t shn_sflow_text_size(const shn_sflow* s, uint32_t g);
int shn_sflow_text(const shn_sflow* s, uint32_t g, uint8_t* out);

/* ---- final containment de-duplication (host, native) ------------------------------------------------
 * Replaces faster_reps.find_reps (faster_reps.py:98-131, duplicate_check_ends :60-92; called `-d` from
 * shannon.py:604).  n FASTA records in file order (names without '>', sequences), r = 24.            */
int shn_find_reps(const uint8_t* names, const uint64_t* name_off, const uint8_t* seqs, const uint64_t* seq_off, uint64_t n,
                  int ds, int r, uint8_t* keep_out);

/* ---- sparse-flow node decomposition ------------------------------------------------------------
 * Replaces the randomized trial loop of path_decompose (path_decompose_sparse.py:100-117): the
 * <=100 cvxopt.solvers.lp calls per decomposed node (cvxopt: third party, version unpinned, not
 * vendored).  cvxopt.solvers.lp is an interior-point method; what is restated is its limit: the flows
 * on the unsupported cells from an exact vertex of the trial LP (they are the same all over the
 * optimal face), the flows on the supported (zero-cost) cells at the ANALYTIC CENTRE of the optimal
 * face (oracle/lp.py: transport_vertex + face_center, bit for bit).  Problem p is the m x n
 * transportation problem with balanced+scaled marginals ab = [a_s (m), b_s (n)] and byte mask
 * mask[j*m+i] = 1 where NO known path supports cell (i,j); trial t draws cost numerators from the
 * counter-based stream (seed, pid[p], t, cell).  flows_out receives, per problem, trials*m*n
 * doubles laid out [cell j*m+i][trial] (problems concatenated).  Thresholding / trial selection
 * (:119-192) stay on the host (shannon_amd/sparse_flow.py, csrc/sflow_host.hip).               */
int shn_lp_solve_batch(shn_ctx* ctx, uint32_t n_problems, const uint32_t* m, const uint32_t* n, const uint32_t* trials,
                       const uint64_t* pid, const double* ab, const uint8_t* mask, uint64_t seed, double* flows_out);
/* The rule of the trial LPs on a context: SHN_LP_RULE_CENTER (default; the interior-point limit above) or SHN_LP_RULE_VERTEX
 * (the vertex itself: the rule of rounds 1-2, kept behind this switch; SHN_LP_RULE=vertex in the environment selects it for
 * every new context).                                                                                                          */
#define SHN_LP_RULE_VERTEX 0
#define SHN_LP_RULE_CENTER 1
int shn_lp_set_rule(shn_ctx* ctx, int rule);
/* Census of the LP calls of a context since its creation / the last reset: out8 = { problems (path_decompose calls that reached
 * the trial loop), problems with a degenerate optimal face in at least one trial (the centre differs from the vertex), trials,
 * trials with a degenerate face, Newton steps, classes that did not converge in 100 steps, trials of problems with more than 64
 * rows + columns (vertex kept), the rule in force }.                                                                            */
int shn_lp_stats(shn_ctx* ctx, uint64_t* out8, int reset);

#ifdef __cplusplus
}
#endif
#endif
