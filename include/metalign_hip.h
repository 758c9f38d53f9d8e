/*
 * metalign_hip.h — C ABI of libmetalign_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the two data-parallel stages of Metalign.
 * Every entry point replaces a process/file seam of the reference (cited as
 * path:line under /root/reference); none of them takes a torch type.  All
 * functions return 0 on success and a negative mg_status on failure; the text
 * of the last failure on the calling thread is available from mg_last_error().
 * Nothing is thrown across the ABI.
 *
 * Ownership: the caller owns every host buffer, in and out.  Pointers named
 * d_* are DEVICE pointers (HBM) owned by the caller (mg_dev_malloc, or any
 * other allocator of the same HIP runtime, e.g. a torch tensor's data_ptr()).
 * Opaque handles (mg_sketch, mg_db, mg_profile) own device memory inside the
 * library and are released by their *_free function or by mg_shutdown().
 *
 * Threading: calls are not re-entrant; one host thread drives one device.
 */
#ifndef METALIGN_HIP_H
#define METALIGN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 1

typedef enum mg_status {
  MG_OK = 0,
  MG_ERR_HIP = -1,       /* a HIP runtime call failed (text in mg_last_error) */
  MG_ERR_ARG = -2,       /* invalid argument (k out of range, null pointer, ...) */
  MG_ERR_CAPACITY = -3,  /* caller-provided output buffer too small */
  MG_ERR_STATE = -4,     /* mg_init not called / handle used out of order */
  MG_ERR_NOMEM = -5
} mg_status;

/* Largest supported k (2-bit packed k-mer in two 64-bit words).  The stock
 * reference uses k=60 with CMash k-range 30-60-10 (scripts/select_db.py:44,50,75);
 * BASELINE.json configs use k in {21,31,51}. */
#define MG_MAX_K 64

/* ------------------------------------------------------------------------ *
 * Alignment record: one retained SAM line, pre-tokenised at ingest.
 * "Retained" = survives the line filter of map_and_process
 * (scripts/map_and_profile.py:206-213: not '@', >= 6 fields, not unmapped,
 * CIGAR != '*').  Field meaning follows the columns the reference reads:
 *   FLAG  [1] :104-111   RNAME [2] :217   CIGAR [5] :86-100   SEQ [9] :142-144
 * ------------------------------------------------------------------------ */
typedef struct mg_aln_rec {
  uint32_t ref_new;   /* bit 31: QNAME differs from the previous retained line
                         (the `read != prev_read` test, :220);
                         bits 0..30: accession index (row of ref2tax)        */
  uint32_t matched;   /* sum of CIGAR 'M' op lengths (filter_line, :91-92)     */
  uint32_t total;     /* sum of all CIGAR op lengths (:94); never 0           */
  uint32_t flag_len;  /* bits 0..11: SAM FLAG; bits 12..31: len(SEQ), 0 if '*' */
} mg_aln_rec;

#define MG_REC_NEW_BIT 0x80000000u
#define MG_REC_REF_MASK 0x7fffffffu
#define MG_REC_FLAG_MASK 0xfffu
#define MG_REC_LEN_SHIFT 12
#define MG_REC_MAX_SEQLEN ((1u << 20) - 1u)

/* ------------------------------------------------------------------------ *
 * Lifecycle, errors, raw device memory
 * ------------------------------------------------------------------------ */
int mg_abi_version(void);
int mg_device_count(void);
/* Test and diagnostic knobs.  The library reads NO environment variable: what rounds 1-4 steered through MG_DEBUG_* / MG_STREAM_* /
 * MG_PGZIP_* variables is set here (key = the variable's name without its prefix, lower case; value 0 = off / the default).
 * key = NULL: every knob back to its default.  Keys: k3_hashed, k3_flush_tiles, k3_grid (stage C: hashed bins on a small taxonomy,
 * flush interval, grid size), lds_pad, force_list, distinct_hint_ppm (stage A's distinct-count estimate as parts per million of the
 * expected candidates: forces table overflows), resident_scan, no_fused, resident_ablate, flush_order (1: filter words first, 2: slots
 * first), no_avx2, gzip_threads, pgzip_chunk, pgzip_thp, pgzip_timing (the host inflater), stream_thin, stream_threads (the file
 * readers), inflate_trace, inflate_loose_find (the device inflater: a line per stage and hole on stderr; block starts by the format's
 * rules alone), inflate_dev_max_bytes (a `.gz` file above this many bytes — or above half of the free device memory — goes through
 * the host inflater, whose memory is bounded by its pieces; the device inflater holds the whole compressed file), shares_threads (mg_multimapped_shares: 1 = the serial loop, 2 / 4 / 8 host threads; 0 = by the list's length),
 * kc_wg_per_cu, kc_ablate (k_count_kmers: workgroups per CU; measurements only — 1: the minimizer runs are dropped, 2: ... after the
 *   gate, 3: no entry is matched, 5: nothing is counted, 6: the tiles are staged and nothing else, 7: ... staged from 8 KB that
 *   stay in the caches, the runs dropped); kc_gate_extra (mg_refdb_index_kmers: the gate has 2^this bits per k-mer; default 6).
 * Needs no device and no mg_init.  MG_ERR_ARG for a key that does not exist. */
int mg_debug_set(const char* key, int64_t value);
int64_t mg_debug_get(const char* key);

/* Binds the calling process to `device` and creates the library stream. */
int mg_init(int device);
/* As mg_init, but launches on a caller-owned hipStream_t, so that the library's kernels are ordered with the
 * caller's own work on that stream (collectives included).  The stream must be an explicit one: the legacy default
 * stream (handle 0, which is what torch.cuda.current_stream() is unless a stream was set) is refused. */
int mg_init_on_stream(int device, void* hip_stream);
void mg_shutdown(void);
const char* mg_last_error(void);
int mg_device_name(char* buf, int cap);
/* Free and total bytes of the device's memory as the runtime reports them (hipMemGetInfo) and the bytes the library's
 * caching allocator holds for reuse — what a caller sizes a batch against (the whole-file ingest of
 * metalign_amd/map_and_profile.py falls back to chunks when the text does not fit) and what tests/ and tools/soak.py
 * watch for growth over thousands of passes.  No counterpart in the reference (host memory is Python's). */
int mg_mem_info(uint64_t* free_bytes, uint64_t* total_bytes, uint64_t* pooled_bytes);
/* Gives the blocks the caching allocator holds for reuse, and the library's grow-only scratch buffers (counting tables
 * among them), back to the runtime (after waiting for the device; they are allocated again on demand).  Worth
 * calling once after a first, worst-case-sized pass: the first stage-A pass of a k has no distinct-count ratio yet and
 * sizes its tables for the worst case (tens of GB against a dense table), which the allocator would otherwise keep. */
int mg_mem_trim(void);

int mg_dev_malloc(void** d_ptr, uint64_t bytes);
int mg_dev_free(void* d_ptr);
int mg_memcpy_h2d(void* d_dst, const void* h_src, uint64_t bytes);
int mg_memcpy_d2h(void* h_dst, const void* d_src, uint64_t bytes);
/* Asynchronous on the library stream (accumulator reset between batches). */
int mg_dev_memset(void* d_ptr, int byte_value, uint64_t bytes);
/* A marker on the library's main stream: mg_event_synchronize returns when everything queued there before
 * mg_event_record has finished — unlike mg_sync, not what was queued after it, so a caller can queue the next
 * batch before it reads the current one back. */
int mg_event_create(void** ev);
int mg_event_record(void* ev);
int mg_event_synchronize(void* ev);
int mg_event_destroy(void* ev);
/* Page-locked host memory and an asynchronous device-to-host copy into it (ordered on the library stream;
 * the bytes are valid after mg_sync).  Lets a caller queue all of a batch's small read-backs behind the
 * kernels and pay for one synchronisation. */
int mg_host_alloc(void** h_ptr, uint64_t bytes);
int mg_host_free(void* h_ptr);
int mg_memcpy_d2h_async(void* h_pinned_dst, const void* d_src, uint64_t bytes);
/* The other direction, from page-locked host memory (which must not be rewritten before the copy has run). */
int mg_memcpy_h2d_async(void* d_dst, const void* h_pinned_src, uint64_t bytes);
int mg_sync(void);
/* Stage C on a second stream of the library (on != 0): its latency-bound pass then overlaps the small kernels
 * that finish stage A and run stage B instead of queueing behind them.  While enabled, every mg_profile_* launch
 * goes to that stream; its inputs must be complete when the call is made (they are after any mg_sync / synchronous
 * copy).  mg_stage_c_join makes the main stream wait for what stage C has queued so far (before reading its
 * accumulators with mg_memcpy_d2h_async); mg_sync waits for both streams. */
int mg_stage_c_side_stream(int on);
/* Stage A on its own stream (on = 1 or 2: which of two such streams the NEXT sketch goes to; 0 = back to the main
 * stream, after waiting for both): mg_sketch_reads_dev_async then queues its whole pipeline (table clear,
 * k_sketch_reads, bucket sort / pack) there, and a caller that processes batch after batch can start the NEXT
 * batch's stage A before it finishes the current batch (stage B, exchange, read-backs on the main stream): the
 * GPU stays on the dominant kernel.  Alternating 1 / 2 between consecutive batches also lets batch i+1's
 * k_sketch_reads overlap batch i's sort / pack tail.  Consumers of such a sketch (mg_containment_dev,
 * mg_sketch_split, mg_sketch_device_ptrs, mg_sketch_download) make the main stream wait for it on the device;
 * mg_sketch_resolve waits for that sketch only.  mg_sync does NOT wait for the stage-A streams.
 * 3 / 4: streams 1 / 2 at the device's LOWEST stream priority (made anew when the kind changes): for a caller whose other streams
 * carry only short kernels — those then get their wavefronts ahead of the next batch's persistent stage-A kernel (a single shard's
 * passes: 5 % faster); with collectives on the main stream the default priority is the faster one. */
int mg_stage_a_side_stream(int on);
/* Resident k_sketch_reads workgroups per CU while on a stage-A stream: 0 = as many as LDS allows, default 2 (leaves
 * issue slots to the small dependent kernels of a multi-GPU exchange running beside it). */
int mg_stage_a_workgroups_per_cu(int n);
int mg_stage_c_join(void);

/* Per-kernel timing with HIP events on the library stream (bench.py's
 * roofline leg).  Names are the kernel family names listed in DESIGN.md. */
int mg_prof_enable(int on);
/* Restrict timing to one kernel family ("" = all); reset by mg_prof_enable. */
int mg_prof_only(const char* kernel);
int mg_prof_reset(void);
/* Returns number of launches and their summed device time in milliseconds. */
int mg_prof_get(const char* kernel, uint64_t* launches, double* total_ms);

/* ------------------------------------------------------------------------ *
 * Stage A — read sketch.
 * Replaces: `kmc -k60 -ci2 -cs3 ...` canonical k-mer counting of the reads
 * (scripts/select_db.py:50-52) and the k-mer enumeration / hashing inside
 * CMash's StreamingQueryDNADatabase.py (scripts/select_db.py:73-76).
 *
 * Definition (the oracle, oracle/mg_oracle.c, is the normative statement):
 * every window of k consecutive [ACGTacgt] bases of every read is upper-cased,
 * replaced by the lexicographically smaller of itself and its reverse
 * complement, and hashed with MurmurHash3_x64_128(ASCII k-mer, seed 0), first
 * 64 bits.  The sketch is the ascending list of DISTINCT hashes <= hmax with
 * their occurrence counts, truncated to the s smallest when s > 0 (`truncated`
 * reports whether entries were cut).  Counts SATURATE at cs = 3 by default,
 * KMC's `-cs3` (scripts/select_db.py:50): count = min(occurrences, cs).  The
 * only thing downstream reads from a count is `>= ci` with ci = 2 (`-ci2`), and
 * merging per-GPU sketches keeps it exact: min(sum of min(c_r, cs), cs) =
 * min(sum of c_r, cs).  mg_set_count_saturation(0) selects exact counts
 * (saturating at 2^32-1); it applies to sketches built afterwards.
 * ------------------------------------------------------------------------ */
typedef struct mg_sketch mg_sketch;
typedef struct mg_filter mg_filter;
int mg_set_count_saturation(uint32_t cs);
uint32_t mg_count_saturation(void);
/* Which definition of a k-mer's hash stage A / A' compute (sketches and tables of different modes do not mix; the table
 * on disk records its mode, metalign_amd/formats.py):
 *   0  MurmurHash3_x64_128(the lexicographically smaller of k-mer and reverse complement)[0:64] — KMC's canonical k-mer,
 *      one hash, the full 64 bits.  Default; what every figure of this build is measured with.
 *   1  min(MurmurHash3(k-mer), MurmurHash3(reverse complement)) mod 9999999999971 — CMash's MinHash.CountEstimator as
 *      SURVEY.md §8(c) recollects it (two hashes per k-mer).  UNVERIFIED: CMash is not under /root/reference and the
 *      reference pins no version and holds no vectors at this seam (scripts/select_db.py:69-76,
 *      local_tests/dump_kmers.py:2-7); the mode exists so that a table built here and one built by CMash could be
 *      compared at all. */
int mg_set_hash_mode(int mode);
int mg_hash_mode(void);
const char* mg_hash_mode1_ks(void); /* "1, 5, 10, ...": the k mode 1 is built for (any other k in that mode: MG_ERR_ARG at the launch) */

int mg_sketch_reads_dev(const uint8_t* d_bases, const uint64_t* d_offsets,
                        uint64_t nreads, int k, uint64_t hmax, uint64_t s,
                        mg_sketch** out);
/* As mg_sketch_reads_dev, without the host synchronisation at the end: the sketch's size, last hash and
 * truncation flag stay on the device until the first call that needs them on the host (any accessor, download,
 * split, set_bound, or mg_sketch_resolve), so a batch can queue stage B behind stage A and synchronise once.
 * mg_containment_dev accepts such a sketch as it is when s == 0.  d_bases / d_offsets must stay valid until the
 * sketch is resolved.  mg_sketch_resolve: *rebuilt = 1 when the sketch had to be recomputed (the counting table
 * inside stage A overflowed): results derived from it before that are stale and must be recomputed. */
int mg_sketch_reads_dev_async(const uint8_t* d_bases, const uint64_t* d_offsets,
                              uint64_t nreads, int k, uint64_t hmax, uint64_t s,
                              mg_sketch** out);
int mg_sketch_resolve(mg_sketch* sk, int* rebuilt);
/* Every k of a multi-k query from ONE pass over the reads — the reference's containment query is multi-k (CMash
 * k-range `30-60-10`, scripts/select_db.py:75; BASELINE configs use {21,31,51}).  ks[nk] ascending, hmaxs[nk] the
 * per-k thresholds, filters[nk] the per-k membership pre-filters (NULL, or NULL entries, = unfiltered).  out[i]
 * receives the sketch of ks[i], pending like one from mg_sketch_reads_dev_async and bit-identical to it.  For the
 * k sets {21,31,51} and {30,40,50,60} a fused kernel stages and rolls the reads once and derives every smaller k's
 * k-mer from the window at the largest k; any other set (or a k whose candidates are few enough for the list path)
 * is served by one launch per k. */
int mg_sketch_reads_multi_dev_async(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, int nk,
                                    const int* ks, const uint64_t* hmaxs, uint64_t s,
                                    const mg_filter* const* filters, mg_sketch** out);
/* A sample that arrives in PIECES — a reads file streamed through page-locked chunks while the next chunk is in
 * flight, or a file larger than the device: every piece is hashed into the SAME per-k counting tables (the tables are
 * what dedupes and counts), so nothing is sketched per piece and nothing merged.  Replaces kmc reading the whole reads
 * file, scripts/select_db.py:45-52 (`.gz` included, :146-148).
 *   begin:   ks[nk] ascending (1..4 of them), thresholds, bottom-s and pre-filters as for mg_sketch_reads_multi_dev_async;
 *            expect_bases = the sample's total bases, roughly (sizes the tables together with the library's
 *            distinct-count hint; an estimate that proves too small is reported at resolution, see below).
 *   add_dev: one batch of reads already in HBM (bases + offsets[nreads + 1]; nbases = their total, 0 = unknown).
 *            Asynchronous; the batch may be freed as soon as the call returns (stream-ordered).
 *   finish:  out[i] = the sketch of ks[i], pending like one from mg_sketch_reads_dev_async and bit-identical to the
 *            sketch of the concatenated batches.  mg_sketch_resolve returns MG_ERR_CAPACITY when a table overflowed
 *            (the reads are gone: the hint is reset to the worst case, stream the sample again).
 *   add_file / add_gzip: the file -> HBM -> parse -> add pipeline inside the library (mg_stream.hip): reader threads
 *            fill page-locked chunks (plain files: positional reads in parallel; gzip: zlib inflate, BGZF blocks in
 *            parallel), one DMA stream uploads chunk i + 1 while chunk i is parsed on the device (mg_reads_parse
 *            rules; record-aligned: the incomplete last record of a chunk is carried to the front of the next ON THE
 *            DEVICE) and hashed into the tables.  format as mg_reads_parse; offset / length: a byte range of the
 *            file that begins on a record boundary (a rank's share; length 0 = to the end).  chunk_bytes / nthreads:
 *            0 = defaults (32 MB, up to 8 readers). */
typedef struct mg_sketch_stream mg_sketch_stream;
int mg_sketch_stream_begin(int nk, const int* ks, const uint64_t* hmaxs, uint64_t s, const mg_filter* const* filters,
                           uint64_t expect_bases, mg_sketch_stream** out);
/* mg_sketch_stream_begin_counts: the same stream of pieces COUNTED by k-mer identity into `kc` against `db`'s k-mer index
 * (mg_count_kmers_dev per piece; mg_sketch_stream_finish then has nothing to return: nk = 0) — what lets
 * mg_sketch_stream_add_file feed stage A by k-mer identity from a reads file.  db and kc outlive the stream. */
struct mg_refdb;
struct mg_kcounts;
int mg_sketch_stream_begin_counts(const struct mg_refdb* db, struct mg_kcounts* kc, mg_sketch_stream** out);
int mg_sketch_stream_add_dev(mg_sketch_stream* ss, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads,
                             uint64_t nbases);
int mg_sketch_stream_add_file(mg_sketch_stream* ss, const char* path, int format, uint64_t offset, uint64_t length,
                              uint64_t chunk_bytes, int nthreads);
int mg_sketch_stream_finish(mg_sketch_stream* ss, mg_sketch** out);
uint64_t mg_sketch_stream_nreads(const mg_sketch_stream* ss);
uint64_t mg_sketch_stream_nbases(const mg_sketch_stream* ss);
void mg_sketch_stream_free(mg_sketch_stream* ss);
/* ------------------------------------------------------------------------ *
 * Membership pre-filter over the genome table's hashes — the role of the bloom pre-filter the reference hands
 * to CMash (`-f cmash_db_n1000_k60_30-60-10.bf`, scripts/select_db.py:70,75).  One bit per hash: bit (h mod
 * 2^b), 2^b = the power of two >= 16 x the number of hashes (2^16 <= 2^b <= 2^30: at most 128 MB, inside the
 * Infinity Cache).  No false negatives, ~6 % false positives (more for tables beyond 64 M hashes).  The FILTERED read sketch is the sketch defined above restricted to the hashes whose bit is
 * set (truncation to s applies after the filter): every table hash stays in, so containment is unchanged, and a
 * table whose largest hash does not filter much (tiny genomes) no longer turns every read k-mer into a table
 * insert.  Build it from ALL hashes of the table (every k has its own), also on a rank that holds a slice.
 * The filter must outlive the sketches built with it until they are resolved.
 * ------------------------------------------------------------------------ */
int mg_filter_build(const uint64_t* hashes, uint64_t n, mg_filter** out);
/* The filter's bit array to the host (2^log2_bits / 8 bytes) and back: the table builder stores it next to the table
 * (the reference ships its pre-filter as a file too, ..._30-60-10.bf, scripts/select_db.py:70), so that a rank that
 * loads 1/W of the table does not have to stream the other W-1 parts just to set their bits. */
int mg_filter_download(const mg_filter* f, uint32_t* bits, uint64_t nbytes);
int mg_filter_from_bits(const uint32_t* bits, unsigned log2_bits, mg_filter** out);
unsigned mg_filter_log2_bits(const mg_filter* f);
/* The RESIDENT INDEX of a table whose largest hash filters little (genomes of a few kb, or viruses beside bacteria: the
 * bloom pre-filter's other job, scripts/select_db.py:70,75): a counting table seeded ONCE with every hash of the table
 * (hashes[0..n), duplicates welcome, all <= hmax) and never cleared — counters carry the epoch of the sketch call that
 * wrote them.  Sketch calls given this filter then cost a candidate ONE random 16-byte access (found: counted; not
 * found: not a hash of the table, dropped) instead of a home slot plus a filter word, and no table clear; the sketch
 * holds exactly the read k-mers that are hashes of the table (the bit filter lets ~6 % of the others through), so
 * containment is unchanged.  16 bytes x 2 to 4 slots per hash (x 2^spread, spread in [0,3]: at a lower load fewer
 * candidates have to look a slot further), once per stream that sketches with it.
 * MG_ERR_CAPACITY: the hashes crowd some range (a bucket without a free slot) — the filter stays a bit filter. */
int mg_filter_make_resident(mg_filter* f, const uint64_t* hashes, uint64_t n, uint64_t hmax, unsigned spread);
int mg_filter_drop_resident(mg_filter* f);   /* back to a bit filter (sketches made with it must have been resolved) */
/* on = 0: sketch calls given this filter take its bit array although it has a resident index (which stays seeded); on != 0:
 * back to the index.  What a job uses to MEASURE both forms on its own sample before it keeps one (the index pays when most
 * candidates are hashes of the table, the bit filter when most are not). */
int mg_filter_use_resident(mg_filter* f, int on);
uint64_t mg_filter_resident_bytes(const mg_filter* f);
void mg_filter_free(mg_filter* f);
int mg_sketch_reads_filtered_dev(const uint8_t* d_bases, const uint64_t* d_offsets,
                                 uint64_t nreads, int k, uint64_t hmax, uint64_t s,
                                 const mg_filter* filter, mg_sketch** out);
int mg_sketch_reads_filtered_dev_async(const uint8_t* d_bases, const uint64_t* d_offsets,
                                       uint64_t nreads, int k, uint64_t hmax, uint64_t s,
                                       const mg_filter* filter, mg_sketch** out);
/* Union of (hash,count) runs, counts of equal hashes summed, then truncated to
 * s: the merge step after an all-gather of per-GPU sketches.  Inputs need not
 * be sorted.  `any_truncated`: OR of the inputs' truncated flags with
 * `bound` = the smallest last-hash among truncated inputs (entries above it
 * are dropped so the union is a complete bottom set). */
int mg_sketch_from_pairs_dev(const uint64_t* d_hashes, const uint32_t* d_counts,
                             uint64_t n, uint64_t s, int any_truncated,
                             uint64_t bound, mg_sketch** out);
/* As mg_sketch_from_pairs_dev, for pairs whose hashes all lie in [range_lo, range_hi] (a hash-range slice):
 * merges through the partitioned counting table instead of a global sort.  Pairs outside the range, or
 * range_hi < range_lo, select the general path. */
int mg_sketch_merge_dev(const uint64_t* d_hashes, const uint32_t* d_counts, uint64_t n,
                        uint64_t range_lo, uint64_t range_hi, uint64_t s, int any_truncated,
                        uint64_t bound, mg_sketch** out);
/* The same merge queued without a host synchronisation: the handle comes back PENDING, like one from
 * mg_sketch_reads_dev_async (mg_containment_dev consumes it on the device; mg_sketch_resolve settles it and reports
 * a redo).  d_hashes / d_counts must stay valid until it is resolved.  Small inputs or inputs without a declared
 * hash range are merged synchronously. */
int mg_sketch_merge_dev_async(const uint64_t* d_hashes, const uint32_t* d_counts, uint64_t n, uint64_t range_lo,
                              uint64_t range_hi, uint64_t s, int any_truncated, uint64_t bound, mg_sketch** out);
/* Positions at which an ascending sketch crosses `nbounds` hash values: out_idx[i] = number of
 * entries with hash < bounds[i].  Used to cut a sketch into hash-range slices for the multi-GPU
 * exchange (rank r owns hashes in [bounds[r-1], bounds[r])). */
int mg_sketch_split(const mg_sketch* sk, const uint64_t* bounds, uint32_t nbounds,
                    uint64_t* out_idx);
/* The device-side form for a multi-GPU exchange that never stops for the host: entries per slice (nbounds + 1 of
 * them) followed by truncated, last hash, n and the counting table's overflow count (non-zero: the sketch will be
 * rebuilt at mg_sketch_resolve and these words are stale), as int64 words at d_out[0 .. nbounds + 4].  d_bounds:
 * ascending hash bounds in device memory.  Works on a sketch whose finalisation is still deferred. */
int mg_sketch_slice_words_dev(const mg_sketch* sk, const uint64_t* d_bounds, uint32_t nbounds, int64_t* d_out);
/* Overrides the completeness bound mg_containment_dev uses for this sketch (a hash-range slice of a
 * truncated sample sketch is complete up to the SAMPLE's last hash, not its own): hashes above
 * `bound` are outside the sketch when `truncated` != 0. */
int mg_sketch_set_bound(mg_sketch* sk, int truncated, uint64_t bound);
uint64_t mg_sketch_size(const mg_sketch* sk);
int mg_sketch_truncated(const mg_sketch* sk);
uint64_t mg_sketch_last_hash(const mg_sketch* sk); /* largest hash in the sketch (0 when empty) */
uint64_t mg_sketch_kmers_seen(const mg_sketch* sk); /* valid k-mer windows hashed */
int mg_sketch_device_ptrs(const mg_sketch* sk, const uint64_t** d_hashes,
                          const uint32_t** d_counts);
int mg_sketch_download(const mg_sketch* sk, uint64_t* hashes, uint32_t* counts,
                       uint64_t cap);
void mg_sketch_free(mg_sketch* sk);

/* Host-buffer convenience: upload, sketch, download. */
int mg_sketch_reads(const uint8_t* bases, const uint64_t* offsets,
                    uint64_t nreads, int k, uint64_t hmax, uint64_t s,
                    uint64_t* out_hashes, uint32_t* out_counts, uint64_t out_cap,
                    uint64_t* out_n, int* out_truncated, uint64_t* out_kmers_seen);

/* ------------------------------------------------------------------------ *
 * Ingest on the device (text already in HBM -> what stage A consumes).
 * Replaces the reads parsing inside kmc (scripts/select_db.py:45-52: -fq / -fa).
 * format 0 = FASTQ (4 lines per record), 1 = FASTA with one sequence line per
 * record (malformed records are an error), 2 = FASTA with sequences over any
 * number of lines: a line starting with '>' opens a record, every other line
 * after the first header is stripped of white space at both ends and appended,
 * lines in front of the first header are ignored.
 * Sequences are kept as written (case, N); '\r' before '\n' is dropped.
 * ------------------------------------------------------------------------ */
typedef struct mg_reads mg_reads;
int mg_reads_parse_dev(const uint8_t* d_text, uint64_t nbytes, int format, mg_reads** out);
/* A PIECE of a reads file that begins on a record boundary (final = 0): the whole records in it are parsed and
 * *consumed = the byte where the first incomplete record begins — the caller carries [consumed, nbytes) to the front of
 * the next piece (mg_sketch_stream_add_file does, on the device).  final != 0: mg_reads_parse_dev. */
int mg_reads_parse_prefix_dev(const uint8_t* d_text, uint64_t nbytes, int format, int final, uint64_t* consumed, mg_reads** out);
int mg_reads_parse(const uint8_t* text, uint64_t nbytes, int format, mg_reads** out);
uint64_t mg_reads_count(const mg_reads* r);
uint64_t mg_reads_nbases(const mg_reads* r);
int mg_reads_device_ptrs(const mg_reads* r, const uint8_t** d_bases, const uint64_t** d_offsets);
int mg_reads_download(const mg_reads* r, uint8_t* bases, uint64_t* offsets);
void mg_reads_free(mg_reads* r);

/* ------------------------------------------------------------------------ *
 * A gzip file's text through the library's PARALLEL inflater (mg_pgzip.hip) — the reference takes `.gz` reads as ordinary
 * input (scripts/select_db.py:146-148; zcat of the selected genomes :103-105).  One gzip stream is entered in the middle by
 * many host threads (deflate block starts found speculatively, back-references into the unknown 32 KB window kept symbolic
 * and resolved in order — the pugz / rapidgzip scheme), every member's CRC-32 and length are checked, trailing garbage
 * after the last member is ignored as gzip does.  mg_sketch_stream_add_file / mg_sam_stream_file use the same decoder for
 * `.gz` input that is not BGZF.  Plain host code: needs no device and no mg_init.
 *   open:  nthreads <= 0 = every core (at most 64).
 *   read:  the next bytes of the inflated stream, up to cap; *n < cap only at the end of the stream (0: nothing left).
 *          MG_ERR_ARG with the text in mg_last_error for a corrupt or truncated stream.
 * ------------------------------------------------------------------------ */
typedef struct mg_gunzip mg_gunzip;
int mg_gunzip_open(const char* path, int nthreads, mg_gunzip** out);
int mg_gunzip_read(mg_gunzip* h, uint8_t* dst, uint64_t cap, uint64_t* n);
void mg_gunzip_close(mg_gunzip* h);

/* zcat of many files into one: `zcat <genome>.gz >> cmashed_db.fna` per selected genome (scripts/select_db.py:103-105, exit codes
 * ignored there).  out_path is created / truncated and receives every file's text — all members of it — in the order of paths;
 * files are inflated by nthreads host threads (<= 0: every core, at most 32) and written at their offsets.  A file that cannot be
 * read, is not gzip, is corrupt or ends inside a member contributes NOTHING, gets a `zcat: <path>: <reason>` line on stderr and
 * failed[i] = 1 (failed may be null); the call still succeeds.  Plain host code: needs no device and no mg_init. */
int mg_zcat_files(const char* const* paths, uint64_t nfiles, const char* out_path, int nthreads, uint64_t* bytes_out, uint8_t* failed);

/* ------------------------------------------------------------------------ *
 * gzip / BGZF inflated ON THE DEVICE (mg_inflate.hip): the compressed bytes cross the link, the text is born in HBM — the
 * reference's `.fq.gz` reads (scripts/select_db.py:50-52,146-148 hands them to kmc) and `zcat` of genomes (:101-105).  One
 * wavefront per job: a BGZF block, or a chunk of a gzip stream entered at a deflate block start found on the device, decoded
 * to 16-bit symbols against the unknown 32 KB window, the windows chained and the symbols resolved by further kernels; every
 * member's CRC-32 and ISIZE are checked (CRC computed on the device); trailing garbage after the last member is ignored as
 * gzip does.  mg_sketch_stream_add_file / mg_sam_stream_file take this path for `.gz` input unless mg_inflate_config turns it
 * off (then: the host inflater above).
 *   mg_inflate_dev:   comp[ncomp] = a whole gzip file in host memory -> its text on the device (stream-ordered on the library
 *                     stream; synchronised on return).  MG_ERR_ARG with zlib's wording in mg_last_error for a corrupt or
 *                     truncated stream.
 *   mg_inflate_config: chunk_bytes = compressed bytes per job of a gzip stream (default 32 KB), stage_bytes = compressed bytes
 *                     decoded together (default, and stage_bytes < 0: sized by the device — as many jobs as it holds at once), ratio = symbols reserved per compressed byte (default 10; a job that
 *                     needs more is decoded again), on = whether the streaming entry points use the device inflater, lane_jobs =
 *                     launches of at least this many jobs decode ONE JOB PER LANE (64 serial decoders per wavefront; off by
 *                     default: it pays from ~25 000 jobs in a launch), smaller ones one job per wavefront; 0 (on, lane_jobs: < 0) leaves a setting as it is.
 *   mg_inflate_stats: counters since the last reset (host seconds of the stages' phases, jobs, jobs decoded again).
 * ------------------------------------------------------------------------ */
typedef struct mg_inflated mg_inflated;
typedef struct mg_inflate_counters {
  uint64_t stages, jobs, redone, find_candidates, find_steps;
  uint64_t blocks, batches, windows, symbols_out;               /* deflate blocks, symbol batches, 64-bit windows, bytes produced */
  uint64_t clk_tables, clk_decode, clk_emit, clk_tail;          /* shader clocks summed over the jobs: where a job's time goes */
  uint64_t clk_sub[6];                                          /* (a -DMGI_SUBCLOCKS build only) inside clk_decode: stage refill, lanes' decode, walk, prefix sums, queueing, rest */
  double find_s, decode_s, resolve_s, stage_s;
} mg_inflate_counters;
int mg_inflate_dev(const uint8_t* comp, uint64_t ncomp, mg_inflated** out);
uint64_t mg_inflated_bytes(const mg_inflated* t);
int mg_inflated_download(const mg_inflated* t, uint8_t* dst);
void mg_inflated_free(mg_inflated* t);
int mg_inflate_config(int64_t chunk_bytes, int64_t stage_bytes, int ratio, int on, int64_t lane_jobs);
int mg_inflate_stats(mg_inflate_counters* out, int reset);

/* Diagnostic, host code only.  With the knob stream_thin set (mg_debug_set) mg_sketch_stream_add_file / mg_sam_stream_file THIN a
 * plain FASTQ / SAM file in their reader threads to what the device parsers read (a FASTQ record -> ">", its sequence line; a SAM
 * line with its SEQ field replaced by a mark + len(SEQ) and its QUAL by '*': from the page cache both files go up at the PCIe
 * link's rate, and half / two thirds of their bytes are never looked at on the device).  Off by default: as built the readers'
 * per-byte work costs more than the link saves (mg_stream.hip, ThinSource).  This writes the thinned
 * text to out_path — kind 0 FASTQ, 1 SAM; piece_bytes = the streaming slot size (0: the default); the same checks and errors
 * as the stream (malformed FASTQ record, lines that are not a whole number of records, a record that does not fit a piece:
 * MG_ERR_CAPACITY). */
int mg_stream_thin_file(const char* path, int kind, uint64_t piece_bytes, int nthreads, const char* out_path);

/* ------------------------------------------------------------------------ *
 * Stage A' — genome sketch table (the pre-built DB the hot path consumes).
 * Replaces: CMash MakeStreamingDNADatabase.py -n 1000 -k 60
 * (local_tests/retrain_and_test_metalign.sh:49) and the .h5 / KMC-dump / bloom
 * trio (scripts/select_db.py:44,69-70).  Same k-mer / hash definition as
 * Stage A; genome g's sketch is its n smallest distinct hashes, ascending, at
 * out_hashes[out_offsets[g] .. out_offsets[g+1]).  out_hashes needs
 * ngenomes*n entries, out_offsets ngenomes+1.
 * ------------------------------------------------------------------------ */
int mg_sketch_genomes(const uint8_t* bases, const uint64_t* offsets,
                      uint64_t ngenomes, int k, uint64_t n,
                      uint64_t* out_hashes, uint64_t* out_offsets);

/* The k < kmax table of hash mode 1 (CMash as SURVEY.md §8(c) recollects it, UNVERIFIED: the smaller-k columns are
 * containments of the k-PREFIXES of the sketched kmax-mers): per genome the bottom-n sketch at kmax under mode 1, every
 * sketched kmax-mer oriented as CMash keeps it (the strand with the smaller hash, the reverse complement on a tie), the
 * mode-1 hash of its first k bases; the genome's entry = its distinct keys, ascending.  Same output layout as
 * mg_sketch_genomes; independent of mg_set_hash_mode.  Replaces (in that mode) the prefix tree of
 * MakeStreamingDNADatabase.py (local_tests/retrain_and_test_metalign.sh:49). */
int mg_sketch_genomes_prefix(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int kmax, int k, uint64_t n,
                             uint64_t* out_hashes, uint64_t* out_offsets);

typedef struct mg_db mg_db;
int mg_db_upload(const uint64_t* hashes, const uint64_t* offsets,
                 uint64_t ngenomes, mg_db** out);
/* The same table from its HASH-MAJOR form (metalign_amd/formats.py, table version 2: what mg_db_upload builds on the
 * device is what the builder already wrote to disk): pair_hash[npairs] ascending, pair_gen[npairs] the genome of every
 * pair, gsize[ngenomes] the number of pairs of every genome IN THIS ARRAY — so a rank of a multi-GPU job uploads only
 * the contiguous run of pairs that falls in its hash range, and nothing is sorted at upload.  Replaces the load of
 * data/cmash_db_n1000_k60.h5 + its KMC dump (scripts/select_db.py:44,69-70).  max_hash: of the WHOLE table. */
int mg_db_upload_sorted(const uint64_t* pair_hash, const uint32_t* pair_gen, uint64_t npairs, const uint32_t* gsize,
                        uint64_t ngenomes, uint64_t max_hash, mg_db** out);
uint64_t mg_db_ngenomes(const mg_db* db);
uint64_t mg_db_max_hash(const mg_db* db); /* the hmax to sketch reads with */
void mg_db_free(mg_db* db);

/* ------------------------------------------------------------------------ *
 * Stage B — containment of each genome sketch in the read sketch.
 * Replaces: `kmc_tools simple ... intersect` (scripts/select_db.py:54-56) and
 * the per-genome containment index of StreamingQueryDNADatabase.py
 * (scripts/select_db.py:73-76; CSV consumed at :80-85).
 * For genome g: bound = last hash of the read sketch if it was truncated,
 * else 2^64-1; sizes[g] = #{h in sketch(g): h <= bound};
 * hits[g] = #{h in sketch(g): h <= bound, h in read sketch with count >= ci}
 * (ci = 2 reproduces kmc -ci2).  The containment index written to the CSV is
 * the double hits/sizes, computed on the host.
 * ------------------------------------------------------------------------ */
int mg_containment_dev(const mg_sketch* q, const mg_db* db, uint32_t ci,
                       uint32_t* d_hits, uint32_t* d_sizes);
/* The same for every k of a pass in ONE launch of each of the three kernels (bucket index, pairs, reduction) — at 10k
 * genomes a launch is a few tens of microseconds of mostly latency, and a three-k pass had nine of them.  qs[i] against
 * dbs[i] into d_hits[i] / d_sizes[i], i < nk <= 4. */
int mg_containment_multi_dev(int nk, const mg_sketch* const* qs, const mg_db* const* dbs, uint32_t ci, uint32_t* const* d_hits,
                             uint32_t* const* d_sizes);
int mg_containment(const uint64_t* q_hashes, const uint32_t* q_counts,
                   uint64_t qn, int q_truncated, uint32_t ci,
                   const uint64_t* db_hashes, const uint64_t* db_offsets,
                   uint64_t ngenomes, uint32_t* out_hits, uint32_t* out_sizes);

/* ------------------------------------------------------------------------ *
 * THE REFERENCE PIPELINE — stages A' / B wired the way scripts/select_db.py wires KMC and CMash.
 * Replaces, together: `kmc -k60 -ci2 -cs3` over the reads (:50-52), `kmc_tools simple <db dump> <reads> intersect` (:54-56),
 * kmc_dump + the FASTA rewrite (:58-65) and StreamingQueryDNADatabase.py <60-mers.fa> <db.h5> <csv> 30-60-10 (:73-76).
 *
 * The reference counts ONLY k_max-mers of the reads, keeps those that occur >= ci times AND are k_max-mers of some genome
 * sketch, and hands that set to the streaming query, which derives the column of every k of its range from the
 * k-PREFIXES of those k_max-mers and of their reverse complements, looked up among the sketched k_max-mers [CMash: upstream
 * recollection, SURVEY.md §8c — parity unpinned like the rest of stage A/B; normative statement: oracle/mg_oracle.c,
 * mgo_refpipe_*; a second statement on strings: tests/indep_sketch.py].  So the read side is stage A at ONE k
 * (mg_sketch_reads_* / mg_sketch_stream_* with k = k_max and the table's threshold / pre-filter), and every smaller k's
 * column is a function of WHICH sketched k_max-mers matched:
 *   containment_k(g) = #{distinct k-prefixes of g's sketched k_max-mers that are the k-prefix of a MATCHED sketched
 *                        k_max-mer (of any genome) or of its reverse complement}
 *                      / #{distinct k-prefixes of g's sketched k_max-mers};
 *   containment_kmax(g) = mg_containment_dev's.
 * A k_max-mer "matches" when its hash (the mode in force, mg_set_hash_mode) is in the read sketch with count >= ci.
 *
 * mg_sketch_genomes_kmers: mg_sketch_genomes plus the k-mer of every sketch entry as the table keeps it (CMash's database
 *   holds the sketches' k-mers beside their hashes: local_tests/dump_kmers.py:7-14), 2-bit packed, first base most
 *   significant, right-aligned in (hi, lo): mode 0 the lexicographically smaller strand, mode 1 the strand with the smaller
 *   MurmurHash3 (the reverse complement on a tie); of the FIRST window of the genome that has the hash.
 * mg_sketch_genomes_kmers_forward (`build_db --sketch_hash forward`): CMash's TRAINING without reverse complements, as
 *   recollected and unverifiable here: a genome's entries are its k-mers with the n smallest distinct
 *   MurmurHash3(k-mer as it stands in the genome) mod 9999999999971, kept as they stand (first window of a value);
 *   out_hashes = what each entry MATCHES by — its hash under the mode in force (canonical k-mer / min of the two strands) —
 *   in the order of the selecting hashes: not ascending, and a genome that holds a k-mer and its reverse complement carries
 *   the same value twice (two k-mers of the sketch, both counted).  For the k of hash mode 1's list.
 * mg_refdb_build: from those genome-major entries of the LARGEST k (ks[nk-1]; ks ascending, nk <= 4), on the device: the
 *   hash-major pairs of k_max and, per k below it, pa / pb per pair (the number — rank among the table's distinct
 *   k-prefixes — of the kept k-mer's k-prefix / of its reverse complement's, 0xffffffff when that is no prefix of the
 *   table) and the count list (distinct (prefix number, genome), ascending) with gsize_k[g] = entries of genome g.
 * mg_refdb_upload_begin: mg_refdb_upload that RETURNS WHILE THE ARRAYS GO UP (reader threads + a DMA thread of the library fill
 *   page-locked slots of their own): the caller's arrays must stay where they are until the first call that reads the table —
 *   any stage-B call, a download, mg_refdb_kmax_table — which waits for them and checks them as mg_refdb_upload does at once
 *   (a corrupt table is reported THERE).  What select_main does while the reads stream (mg_refdb_max_hash, _ngenomes, _nk
 *   answer at once).
 * mg_refdb_upload: the same handle from stored arrays (metalign_amd/formats.py, table version 3) — or from a rank's share:
 *   any contiguous run of the pairs with its pa / pb, any contiguous run of a count list with gsize counted within it;
 *   nprefix is always the whole table's.  mg_refdb_download_*: what the builder stores.
 * ------------------------------------------------------------------------ */
typedef struct mg_refdb mg_refdb;
int mg_sketch_genomes_kmers(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int k, uint64_t n,
                            uint64_t* out_hashes, uint64_t* out_kmer_hi, uint64_t* out_kmer_lo, uint64_t* out_offsets);
int mg_sketch_genomes_kmers_forward(const uint8_t* bases, const uint64_t* offsets, uint64_t ngenomes, int k, uint64_t n,
                                    uint64_t* out_hashes, uint64_t* out_kmer_hi, uint64_t* out_kmer_lo, uint64_t* out_offsets);
int mg_refdb_build(const uint64_t* hashes, const uint64_t* kmer_hi, const uint64_t* kmer_lo, const uint64_t* offsets,
                   uint64_t ngenomes, int nk, const int* ks, mg_refdb** out);
int mg_refdb_upload(uint64_t ngenomes, int nk, const int* ks, uint64_t npairs, const uint64_t* pair_hash,
                    const uint32_t* pair_gen, const uint32_t* gsize_kmax, uint64_t max_hash, const uint32_t* const* pa,
                    const uint32_t* const* pb, const uint64_t* nprefix, const uint32_t* const* cid,
                    const uint32_t* const* cgen, const uint64_t* ncount, const uint32_t* const* gsize, mg_refdb** out);
int mg_refdb_upload_begin(uint64_t ngenomes, int nk, const int* ks, uint64_t npairs, const uint64_t* pair_hash,
                    const uint32_t* pair_gen, const uint32_t* gsize_kmax, uint64_t max_hash, const uint32_t* const* pa,
                    const uint32_t* const* pb, const uint64_t* nprefix, const uint32_t* const* cid,
                    const uint32_t* const* cgen, const uint64_t* ncount, const uint32_t* const* gsize, mg_refdb** out);
/* npairs; nprefix[nk-1], ncount[nk-1] (any may be NULL) */
int mg_refdb_sizes(const mg_refdb* db, uint64_t* npairs, uint64_t* nprefix, uint64_t* ncount);
/* any pointer may be NULL; kmer_hi / kmer_lo (pair order) only from a table built here */
int mg_refdb_download_kmax(const mg_refdb* db, uint64_t* pair_hash, uint32_t* pair_gen, uint32_t* gsize, uint64_t* kmer_hi,
                           uint64_t* kmer_lo);
int mg_refdb_download_k(const mg_refdb* db, int ki, uint32_t* pa, uint32_t* pb, uint32_t* cid, uint32_t* cgen,
                        uint32_t* gsize);
int mg_refdb_nk(const mg_refdb* db);
uint64_t mg_refdb_ngenomes(const mg_refdb* db);
uint64_t mg_refdb_max_hash(const mg_refdb* db); /* the hmax to sketch the reads' k_max-mers with */
/* the table of the largest k as a plain mg_db (owned by db): for mg_containment_dev and the table-wide helpers */
const mg_db* mg_refdb_kmax_table(const mg_refdb* db);
void mg_refdb_free(mg_refdb* db);
/* Stage B of the reference pipeline.  q: the read sketch of the table's LARGEST k (complete: s = 0; a pending sketch is
 * consumed on the device like mg_containment_dev does).  d_hits[i] / d_sizes[i]: the column of ks[i], ngenomes u32 each.
 * Three launches for the largest k (bucket index, pairs — a matched pair also sets the bits of its prefixes —, reduction)
 * and three for all smaller k together (the count lists streamed against the prefix bitmaps). */
int mg_refpipe_containment_dev(const mg_sketch* q, const mg_refdb* db, uint32_t ci, uint32_t* const* d_hits,
                               uint32_t* const* d_sizes);
/* The two halves, for a multi-GPU job whose ranks hold hash-range slices of the pairs and prefix-range slices of the count
 * lists: mark = the largest k's column of this rank's pairs + this rank's prefix bitmaps (zeroed, then set; mg_refdb_marks
 * gives bitmap ki's device words for the exchange — the bitwise OR over the ranks is what count needs); count = the smaller
 * k's columns (nk - 1 of them) from bitmaps d_marks[ki] (NULL: the handle's own). */
int mg_refpipe_mark_dev(const mg_sketch* q, const mg_refdb* db, uint32_t ci, uint32_t* d_hits_kmax, uint32_t* d_sizes_kmax);
int mg_refpipe_count_dev(const mg_refdb* db, const uint32_t* const* d_marks, uint32_t* const* d_hits,
                         uint32_t* const* d_sizes);
int mg_refdb_marks(const mg_refdb* db, int ki, uint32_t** d_marks, uint64_t* nwords);

/* ------------------------------------------------------------------------ *
 * Stage A of the reference pipeline BY K-MER IDENTITY (round 6; the default of the reference-pipeline path).
 * Replaces `kmc -k<kmax> -ci2 -cs3` over the reads + `kmc_tools simple ... intersect` with the sketches' k-mers
 * (scripts/select_db.py:50-59) as those tools compute it: canonical k-mers compared as k-mers — the read side hashes
 * nothing (MurmurHash3 only SELECTS a genome's sketch, when the table is built).  Normative: oracle/mg_oracle.c,
 * mgo_refpipe_count_kmers.  Differs from the hash path (mg_sketch_* + mg_refpipe_mark_dev) only where two k-mers share a
 * hash value.  k_max in [15, 64]; other k stay with the hash path (and below k_max = 25 the hash path is the faster of the two:
 * the host code chooses it there unless told otherwise).
 *
 * mg_refdb_index_kmers: builds, on the device, the index the read side needs over the table's distinct canonical k_max-mers
 *   (filed under the hash of the 19 to 31 bases around a k-mer's minimizer, four entries to a 128-byte bucket, a gate bitmap
 *   over the hash's leading bits: metalign_amd/csrc/mg_kcount_core.h; 210 B per k-mer, 7 ms for ten million).  kmer_hi / kmer_lo: the kept k_max-mer of every pair in pair
 *   order, 2-bit packed, first base most significant (table format 3: k<K>.kmer_hi.u64 / .kmer_lo.u64) — or NULL for a table
 *   built here by mg_refdb_build, which holds them.  A rank of a multi-GPU job indexes the WHOLE table's pairs it was given.
 * mg_kcounts: one sample's occurrence counters for one table (zeroed when made; mg_kcounts_reset zeroes again, stream-ordered).
 * mg_count_kmers_dev: adds the k_max-mers of a batch of reads resident in device memory (bases as ASCII, offsets[nreads + 1];
 *   nbases = offsets[nreads] - offsets[0], for sizing only) — one launch on the library's current stream, no host sync; a
 *   sample streamed in pieces is a sequence of these calls.  Windows over non-ACGT symbols are skipped, either case counts.
 * mg_kcounts_stats: [0] k-mers of the reads (KMC's total), [1] minimizer runs, [2] runs past the gate, [3] matched windows.
 * mg_kcounts_download: per PAIR of the table min(occurrences of its k-mer, cs) (cs: mg_set_count_saturation, 0 = exact).
 * mg_kcounts_device: the raw counters (u32[npairs], meaningful at the pairs mg_refdb_kmer_heads names) for a multi-GPU sum.
 * mg_kcounts_pack2_dev / _merge2_dev: that sum for counters that saturate at 3 or below (kmc -cs3, the reference's setting): a
 *   rank packs min(counter, 3) of every pair into two bits (mg_kcounts_pack2_bytes bytes, a multiple of four), the ranks
 *   all-gather the arrays (2.5 MB per ten million pairs and rank), and every rank sets its counters to the sum over the ranks'
 *   arrays (d_all: nranks arrays, stride_dwords apart) — min(sum, cs) is what the sample's reads, counted together, would give.
 *   Streams: mg_count_kmers_dev runs on the stage-A stream when mg_stage_a_side_stream is on (the main stream otherwise),
 *   mg_kcounts_reset always on a stream of its own (zero a set of counters when you are done with it: the copies and fills then run
 *   beside the next sample's counting), every other call here on the main stream; an event per set of counters orders them all.
 * mg_refpipe_mark_counts_dev / _containment_counts_dev: mg_refpipe_mark_dev / _containment_dev with "the pair's k-mer occurred
 *   >= ci times" read from the counters instead of a read sketch (_ptr_: from counters the caller summed over the ranks). */
typedef struct mg_kcounts mg_kcounts;
int mg_refdb_index_kmers(mg_refdb* db, const uint64_t* kmer_hi, const uint64_t* kmer_lo);
int mg_refdb_has_kmer_index(const mg_refdb* db);
uint64_t mg_refdb_distinct_kmers(const mg_refdb* db);
int mg_refdb_kmer_heads(const mg_refdb* db, uint32_t* head); /* u32[npairs]: the pair that counts for each pair's k-mer */
int mg_kcounts_new(const mg_refdb* db, mg_kcounts** out);
int mg_kcounts_reset(mg_kcounts* kc);
int mg_count_kmers_dev(const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads, uint64_t nbases, const mg_refdb* db,
                       mg_kcounts* kc);
int mg_kcounts_stats(const mg_kcounts* kc, uint64_t* out4);
int mg_kcounts_download(const mg_kcounts* kc, const mg_refdb* db, uint32_t* per_pair);
int mg_kcounts_device(const mg_kcounts* kc, uint32_t** d_counts, uint64_t* n);
int mg_kcounts_wait(const mg_kcounts* kc); /* the main stream waits for whatever was queued on the counters last (mg_kcounts_device's users) */
uint64_t mg_kcounts_pack2_bytes(const mg_kcounts* kc);
int mg_kcounts_pack2_dev(const mg_kcounts* kc, uint32_t* d_out);
int mg_kcounts_merge2_dev(mg_kcounts* kc, const uint32_t* d_all, uint32_t nranks, uint64_t stride_dwords);
void mg_kcounts_free(mg_kcounts* kc);
int mg_refpipe_mark_counts_dev(const mg_kcounts* kc, const mg_refdb* db, uint32_t ci, uint32_t* d_hits_kmax, uint32_t* d_sizes_kmax);
int mg_refpipe_mark_counts_ptr_dev(const uint32_t* d_counts, const mg_refdb* db, uint32_t ci, uint32_t* d_hits_kmax,
                                   uint32_t* d_sizes_kmax);
int mg_refpipe_containment_counts_dev(const mg_kcounts* kc, const mg_refdb* db, uint32_t ci, uint32_t* const* d_hits,
                                      uint32_t* const* d_sizes);
/* A rank of a multi-GPU job that holds the WHOLE table (stage A by k-mer identity): from now on the count step of this handle streams
 * only entries [n r / W, n (r + 1) / W) of every smaller k's count list — the columns of k < k_max it writes are this rank's PART of
 * them (the ranks' parts add up: the job's all-reduce), the k_max column and every size stay whole.  (1 of 1: everything, the default.) */
int mg_refdb_set_count_share(const mg_refdb* db, uint32_t rank, uint32_t world);

/* ------------------------------------------------------------------------ *
 * Stage C — per-read taxon assignment + abundance histogram.
 * Replaces the loop of map_and_process (scripts/map_and_profile.py:193-264)
 * with parse_flag :104-111, filter_line :86-100, clean_read_hits :130-147,
 * intersect_read_hits :115-125 and process_read :152-176, INCLUDING the
 * carried state of :229-232 (an Ambiguous read drops the next read's first
 * line), the always-Ambiguous first boundary (:155-156) and the unflushed
 * last read (:259-264).
 *
 * A shard is a contiguous range of records that starts on a read boundary.
 * mg_profile_begin_dev computes the shard's composed state map
 * (incoming "first line dropped" bit -> outgoing bit); the caller composes
 * the maps of preceding shards (first shard: incoming = 1, the phantom
 * boundary) and then calls mg_profile_commit_dev.
 *   has_lookahead != 0: d_recs[nrecs] is the first record of the next shard
 *     (its pair flags decide the last read of this shard, :225-226);
 *   has_lookahead == 0: this is the global last shard; its last read is
 *     never processed (reference behaviour).
 * Outputs (device, zero-initialised by the caller, accumulated into):
 *   d_count[t], d_bases[t]  unique reads / bases per dense taxon id (:235-240)
 *   d_first_seen[t]         min over unique reads of (group_base + local
 *                           read index); caller initialises to UINT64_MAX.
 *                           Rebuilds dict insertion order (:240).
 *   d_scalars[0] += reads (tot_rds, :221); d_scalars[1] += Ambiguous reads
 *   (:229-231, the phantom boundary included when first_shard != 0).
 * Multimapped reads (:245-248) are kept in the handle and fetched with
 * mg_profile_multimapped (CSR, ascending read index).
 * ------------------------------------------------------------------------ */
typedef struct mg_profile mg_profile;

int mg_profile_begin_dev(const mg_aln_rec* d_recs, uint64_t nrecs,
                         int has_lookahead, const uint32_t* d_ref2tax,
                         uint32_t nref, uint32_t ntax, double pct_id,
                         mg_profile** out);
/* Resets the accumulators a commit adds into (count = bases = 0, first_seen = UINT64_MAX, scalars = 0);
 * asynchronous, one launch. */
int mg_profile_acc_reset(uint64_t* d_count, uint64_t* d_bases, uint64_t* d_first_seen,
                         uint64_t* d_scalars, uint32_t ntax);
/* Queues the map-only pass behind whatever is on the stream, without synchronising: mg_profile_state_map /
 * mg_profile_ngroups then only read its two result words back. */
int mg_profile_map_launch(mg_profile* p);
/* map[0] = outgoing bit if incoming is 0, map[1] = ... if incoming is 1. */
int mg_profile_state_map(const mg_profile* p, uint8_t map[2]);
/* The same three numbers (map[0], map[1], reads) written as int64 words to device memory behind the map-only pass,
 * without a synchronisation (scripts/map_and_profile.py:229-232 carried state; see mg_profile_state_map). */
int mg_profile_map_words_dev(mg_profile* p, int64_t* d_out3);
uint64_t mg_profile_ngroups(const mg_profile* p);
int mg_profile_commit_dev(mg_profile* p, int incoming_dropped, int first_shard,
                          uint64_t group_base, uint64_t* d_count,
                          uint64_t* d_bases, uint64_t* d_first_seen,
                          uint64_t* d_scalars);
/* As mg_profile_commit_dev for a batch of its own: the accumulators are reset (as by mg_profile_acc_reset) in the
 * launch that prepares the pass, instead of being added to. */
int mg_profile_commit_reset_dev(mg_profile* p, int incoming_dropped, int first_shard,
                                uint64_t group_base, uint64_t* d_count,
                                uint64_t* d_bases, uint64_t* d_first_seen,
                                uint64_t* d_scalars);
/* Sizes of the multimapped CSR after commit. */
int mg_profile_multimapped_size(const mg_profile* p, uint64_t* nreads,
                                uint64_t* nentries);
/* mm_offsets[nreads+1], mm_tax[nentries] (dense taxon ids, SAM order within
 * the read), mm_hitlen[nreads], mm_read[nreads] (group_base + local index). */
int mg_profile_multimapped(const mg_profile* p, uint64_t* mm_offsets,
                           uint32_t* mm_tax, uint64_t* mm_hitlen,
                           uint64_t* mm_read);
/* resolve_multi_prop (scripts/map_and_profile.py:269-312) on the device, over the multimapped CSR of a committed
 * shard, without bringing the CSR to the host.  d_weight[t] = unique bases of taxon t after the read cutoff
 * (:428), NaN for a taxon that was dropped or never hit uniquely (:180-188); d_genome_len[t] = summed accession
 * lengths when --length_normalize is on, else NULL.  d_extra[t] (ntax doubles, overwritten) receives what the
 * reference adds to taxids2abs[t][1] at :310-311.  Sums are floating point and order-dependent: equal to the
 * host computation to ~1e-15 relative, not bit for bit. */
int mg_profile_resolve_multimapped_dev(const mg_profile* p, const double* d_weight,
                                       const double* d_genome_len, double* d_extra);
/* The same on the HOST, from the CSR mg_profile_multimapped returned, in the reference's order of additions — per read the
 * distinct taxa that still have a weight, ascending, share the read's hitlen in proportion to their weights; a taxon's
 * additions are summed read after read (:290-311) — so the result is the host tail's own, bit for bit (the default tail;
 * the device version above is the option).  weight[t] NaN = no entry; genome_len NULL unless --length_normalize;
 * extra[ntax] and touched[ntax] are overwritten.  Plain host code: needs no device and no mg_init. */
int mg_multimapped_shares(const uint64_t* mm_offsets, uint64_t nreads, const uint32_t* mm_tax, const uint64_t* mm_hitlen,
                          const double* weight, uint32_t ntax, const double* genome_len, double* extra, uint8_t* touched);
void mg_profile_free(mg_profile* p);

/* SAM text in HBM -> alignment records (mg_aln_rec), one per retained line, in file order.
 * Replaces, per line, map_and_process's filter ('@', < 6 fields, unmapped, CIGAR '*':
 * scripts/map_and_profile.py:202-213), parse_flag (:104-111), filter_line's CIGAR walk (:88-95),
 * RNAME -> accession row (:217) and the `read != prev_read` test (:220).
 * mg_acc_index: accession names -> row (= index into ref2tax); a repeated name keeps its last row.
 * prev_qname: QNAME of the last retained line of the previous chunk ("" for the first chunk); chunks
 * must end on line boundaries.  On a line the reference cannot parse the call fails with MG_ERR_ARG and
 * reports err_kind (1 KeyError: unknown RNAME, 2 IndexError: < 12 fields, 3 ValueError: FLAG / CIGAR / tag,
 * 4 ZeroDivisionError: CIGAR without operations, 5 alignment too long for the record) and the 0-based
 * line number within the chunk, so that the wrapper can raise what the reference raises. */
typedef struct mg_acc_index mg_acc_index;
typedef struct mg_sam_batch mg_sam_batch;
int mg_acc_index_build(const char* names, const uint64_t* name_offsets, uint32_t nacc, mg_acc_index** out);
void mg_acc_index_free(mg_acc_index* ix);
int mg_sam_tokenize_dev(const uint8_t* d_text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                        mg_sam_batch** out, int* err_kind, uint64_t* err_line);
int mg_sam_tokenize(const uint8_t* text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                    mg_sam_batch** out, int* err_kind, uint64_t* err_line);
/* PAF replay adaptor (minimap2 PAF instead of SAM).  The reference has no PAF reader — it parses SAM
 * columns (scripts/map_and_profile.py:87,97,142-144,211,217) — so this maps PAF onto the same records: fields split on
 * TAB, lines with fewer than 12 fields skipped, RNAME <- column 6, FLAG <- 16 for strand '-' (+ 256 when the last
 * `tp:A:` tag is S), CIGAR totals from the last `cg:Z:` tag (matched = sum of M, total = all operations + the query
 * bases outside [qstart, qend), which SAM writes as clipping) or, without one, matched = column 10 and total = column 2;
 * len(SEQ) <- query length (0 for secondaries).  Same batch handle, same err_kind / err_line contract as the SAM call. */
int mg_paf_tokenize_dev(const uint8_t* d_text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                        mg_sam_batch** out, int* err_kind, uint64_t* err_line);
int mg_paf_tokenize(const uint8_t* text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname,
                    mg_sam_batch** out, int* err_kind, uint64_t* err_line);
/* The alignment FILE (SAM, paf = 0; PAF, paf = 1; plain, gzip or BGZF) -> records on the device through the pipeline of
 * mg_stream.hip: reader threads fill page-locked chunks, chunk i + 1 goes up while chunk i — cut at its last newline, the
 * rest carried to the front of the next chunk on the device — is tokenised, the previous retained QNAME carried along.
 * The batch equals mg_sam_tokenize_dev's of the whole text.  offset / length: a line-aligned byte range of a plain file
 * (length 0 = to the end); chunk_bytes / nthreads: 0 = defaults.  Replaces the line loop of map_and_process,
 * scripts/map_and_profile.py:201-217.  err_line is relative to the piece the line fell into. */
int mg_sam_stream_file(const char* path, int paf, const mg_acc_index* ix, uint64_t offset, uint64_t length,
                       uint64_t chunk_bytes, int nthreads, mg_sam_batch** out, int* err_kind, uint64_t* err_line);
uint64_t mg_sam_batch_count(const mg_sam_batch* b);
const char* mg_sam_batch_last_qname(const mg_sam_batch* b);
int mg_sam_batch_device_ptr(const mg_sam_batch* b, const mg_aln_rec** d_recs);
int mg_sam_batch_download(const mg_sam_batch* b, mg_aln_rec* recs);
void mg_sam_batch_free(mg_sam_batch* b);

/* Host-buffer convenience, single shard = whole stream. Capacities: mm_* as
 * above with mm_cap_reads / mm_cap_entries entries available. */
int mg_profile_assign(const mg_aln_rec* recs, uint64_t nrecs,
                      const uint32_t* ref2tax, uint32_t nref, uint32_t ntax,
                      double pct_id, uint64_t* out_count, uint64_t* out_bases,
                      uint64_t* out_first_seen, uint64_t* out_tot_rds,
                      uint64_t* out_n_ambig, uint64_t* mm_offsets,
                      uint32_t* mm_tax, uint64_t* mm_hitlen, uint64_t* mm_read,
                      uint64_t mm_cap_reads, uint64_t mm_cap_entries,
                      uint64_t* mm_nreads, uint64_t* mm_nentries);

#ifdef __cplusplus
}
#endif
#endif /* METALIGN_HIP_H */
