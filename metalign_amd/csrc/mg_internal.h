// mg_internal.h — shared state of libmetalign_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/metalign_hip.h"

namespace mg {

constexpr uint64_t kReservedHash = 0xFFFFFFFFFFFFFFFFull;  // never a sketch member

struct ProfEntry {
  uint64_t launches = 0;
  double total_ms = 0.0;
};

struct DevBuf;
}  // namespace mg
struct mg_sketch;
struct mg_filter;
namespace mg {

struct Context {
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // optional second stream for stage C (mg_stage_c_side_stream): its latency-bound pass then overlaps the small
  // kernels that finish stage A and run stage B instead of queueing behind them
  hipStream_t stream_c = nullptr;
  hipEvent_t ev_c = nullptr;
  bool c_side = false;
  // optional stream for stage A (mg_stage_a_side_stream): the heavy sketch pipeline of the NEXT batch then runs
  // while the current batch's stage B / exchange / read-backs proceed on the main stream
  hipStream_t stream_a = nullptr, stream_a2 = nullptr;  // two, so that consecutive batches' stage A can overlap
  // the device inflater's stream (mg_inflate.hip): a stage of a .gz file decodes while the previous stage's text is parsed and hashed
  hipStream_t stream_inf = nullptr;
  hipStream_t stream_r = nullptr;   // where a sample's k-mer counters are zeroed (mg_kcounts_reset): beside everything else
  bool inf_side = false;
  int a_side = 0;                                        // 0 off, 1 / 2 = which of them the next sketch goes to
  bool a_low = false;                                    // ... made at the device's lowest stream priority (mg_stage_a_side_stream(3 / 4))
  unsigned a_side_wg_per_cu = 2;                         // k_sketch_reads workgroups per CU on those streams (0 = LDS limit)
  bool is_stage_a(hipStream_t st) const { return st && (st == stream_a || st == stream_a2); }
  const char* stage_a_prefix(hipStream_t st) const { return st == stream_a2 ? "b:" : "a:"; }
  const char* scratch_prefix = "";  // scratch buffers are per stream ("a:" while launching on stream_a)
  std::vector<hipEvent_t> ev_pool;   // recycled completion events of deferred sketches
  int num_cus = 256;
  // stage C's private overflow bins (mg_profile.hip): which block / partition is known to be all-zero
  void* k3_priv_ptr = nullptr;
  uint64_t k3_priv_nb = 0;
  uint32_t* k3_flags = nullptr;  // ... its overflow flags (zeroed by every k_pass_prepare)
  uint32_t k3_nflags = 0;
  // occurrence counters of read sketches saturate here (kmc -cs3, scripts/select_db.py:50); 0 = exact counts
  uint32_t count_sat = 3;
  // which definition of a k-mer's hash the stage-A / A' kernels compute (mg_set_hash_mode; mg_kmer.h)
  int hash_mode = 0;
  // the last epoch handed to a call that counts in a resident index (mg_filter::Resident): unique across filters and copies
  uint32_t resident_epoch = 0;
  // profiling
  bool prof_on = false;
  char prof_only[32] = "";  // when set, only this kernel family is timed (keeps the timed region light)
  std::map<std::string, ProfEntry> prof;
  struct Pending { std::string name; hipEvent_t a, b; };
  std::vector<Pending> prof_pending;   // recorded, not yet read (no sync inside the timed region)
  std::vector<hipEvent_t> prof_pool;   // recycled events
  // grow-only named scratch buffers (sort temp storage, candidate lists, ...) so that
  // steady-state calls do not hipMalloc/hipFree (hipFree synchronises the device)
  std::map<std::string, DevBuf*> scratch;
  std::map<uint64_t, std::vector<void*>> pool;  // size class -> free blocks
  // blocks freed while side streams were in use wait here until every stream has passed the point of the free
  struct FenceBatch { std::vector<std::pair<void*, uint64_t>> blocks; std::vector<hipEvent_t> events; };
  std::vector<std::pair<void*, uint64_t>> fence_open;
  std::vector<FenceBatch> fence_sealed;
  std::vector<hipEvent_t> fence_events;
  uint64_t* pinned = nullptr;
  // landing words of sketches whose finalisation is deferred (mg_sketch_reads_dev_async): kPendSlots x 8 words
  uint64_t* pend_pinned = nullptr;
  static constexpr unsigned kPendSlots = 64;  // (a pipelined multi-k exchange keeps ~8 per k pending: fronts + merged slices)
  struct ::mg_sketch* pend_owner[kPendSlots] = {};
  unsigned pend_next = 0;
};

Context& ctx();
int fail(int code, const char* fmt, ...);
// A test / diagnostic knob (mg_debug_set; include/metalign_hip.h lists them): 0 when it was never set.  The library reads no
// environment variable.
int64_t dbg(const char* key);

// The stage-A kernels' cs word (mg_sketch_dev.h: kCsMask): count saturation + the tests' flush-order pin.
inline uint32_t stage_a_cs_word() {
  const uint32_t order = (uint32_t)dbg("flush_order") & 3u;  // 1: filter words first, 2: slots first (the tests pin either)
  const uint32_t ablate = (uint32_t)dbg("resident_ablate") & 3u;  // (diagnostics: the one-k kernel's look-ups in a resident index)
  return ctx().count_sat | (order << 30) | (ablate << 28);
}

#define MG_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return ::mg::fail(MG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),   \
                        __FILE__, __LINE__);                                                  \
  } while (0)

#define MG_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != MG_OK) return rc__; \
  } while (0)

#define MG_REQUIRE_READY()                                                        \
  do {                                                                            \
    if (!::mg::ctx().ready) return ::mg::fail(MG_ERR_STATE, "mg_init not called"); \
  } while (0)

// Size-class cached device allocator.  pool_alloc returns nullptr (error text set) on failure.
void* pool_alloc(uint64_t bytes, uint64_t* got);
void pool_free(void* p, uint64_t bytes);
void pool_release_all();
// Pinned host words for small read-backs (index < 64).
uint64_t* host_words();

// RAII device buffer on the library stream's device.
struct DevBuf {
  void* p = nullptr;
  uint64_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  // Allocations come from a size-class cache (pool_alloc / pool_free in mg_core.hip): steady-state
  // batches reuse the previous batch's buffers instead of paying hipMalloc / hipFree (which
  // synchronises the device) per handle.
  int alloc(uint64_t n) {
    release();
    uint64_t got = 0;
    p = pool_alloc(n, &got);
    if (!p) return MG_ERR_NOMEM;
    bytes = got;
    return MG_OK;
  }
  void release() {
    if (p) { pool_free(p, bytes); p = nullptr; bytes = 0; }
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Grow-only scratch buffer `name` of at least `bytes` bytes; nullptr (and error text set) on failure.
void* scratch(const char* name, uint64_t bytes);
void scratch_release_all();
// mg_stream.hip: the page-locked slots, DMA stream and events of the file pipelines (kept between calls).
void stream_release_all();
// mg_inflate.hip: the page-locked slots and copy stream of the compressed-byte uploads.
void inflate_release_all();
// ... host arrays in pageable memory -> the device through those slots (reader threads + one DMA stream); st waits for the last piece.
int upload_ranges(const std::vector<std::pair<const void*, std::pair<void*, uint64_t>>>& ranges, hipStream_t st);
struct UploadJob;
int upload_ranges_begin(const std::vector<std::pair<const void*, std::pair<void*, uint64_t>>>& ranges, UploadJob** out);
int upload_ranges_end(UploadJob* job, hipStream_t st);
void upload_ranges_abort(UploadJob* job);

// Times one kernel family with HIP events on the library stream when profiling is on.
struct ProfScope {
  const char* name;
  bool on;
  hipStream_t stream;
  hipEvent_t a = nullptr, b = nullptr;
  explicit ProfScope(const char* n, hipStream_t st = nullptr);  // nullptr: the library stream
  ~ProfScope();
};
// The stream stage C launches on.
inline hipStream_t stage_c_stream() { return ctx().c_side ? ctx().stream_c : ctx().stream; }
// Reads every pending event pair (synchronises on them) into ctx().prof.
void prof_collect();

inline unsigned grid_for(uint64_t items, unsigned per_block, unsigned max_blocks) {
  uint64_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (unsigned)b;
}

// ---- internal services implemented in mg_sort.hip (rocPRIM-backed plain library ops) ----
// Sorts n u64 keys ascending using bits [0,end_bit). out may not alias in.
int sort_keys(const uint64_t* d_in, uint64_t* d_out, uint64_t n, unsigned end_bit);
int sort_keys_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n);
int sort_pairs(const uint64_t* d_kin, uint64_t* d_kout, const uint32_t* d_vin, uint32_t* d_vout, uint64_t n);
// Run-length encode sorted keys -> (unique, counts, *d_runs); asynchronous, run count stays on the device.
int rle_keys(const uint64_t* d_sorted, uint64_t n, uint64_t* d_unique, uint32_t* d_counts, uint64_t* d_runs);
// Sum values of equal adjacent keys (saturating u32) -> (unique, sums, *d_runs); asynchronous.
int reduce_pairs(const uint64_t* d_keys, const uint32_t* d_vals, uint64_t n, uint64_t* d_unique, uint32_t* d_sums,
                 uint64_t* d_runs);
// Sort every segment [offs[i], offs[i+1]) of keys independently.
int segmented_sort_keys(const uint64_t* d_in, uint64_t* d_out, uint64_t n, const uint64_t* d_offsets, uint64_t nseg);
// Sort (key u64, value u32) pairs by key (all 64 bits); used once per table upload.
int sort_pairs(const uint64_t* d_kin, uint64_t* d_kout, const uint32_t* d_vin, uint32_t* d_vout, uint64_t n);
// Exclusive prefix sums; *h_total receives the grand total.
int exclusive_sum_u32_to_u64(const uint32_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* h_total);

}  // namespace mg

// Opaque handle layouts (shared between translation units).
struct mg_sketch {
  mg::DevBuf hashes;   // u64[n]
  mg::DevBuf counts;   // u32[n]
  uint64_t n = 0;
  int truncated = 0;
  uint64_t kmers_seen = 0;
  uint64_t last_hash = 0;  // hashes[n-1] when n > 0
  bool has_bound = false;  // mg_sketch_set_bound: completeness bound of the sample this slice belongs to
  uint64_t bound = 0;
  // bucket index for containment look-ups (built lazily)
  mg::DevBuf index;    // u32[nbuckets+1]
  unsigned index_shift = 0;
  uint64_t index_buckets = 0;
  // Deferred finalisation (mg_sketch_reads_dev_async): until the first host-side read, n / last_hash /
  // truncated / kmers_seen live in `meta` on the device ([0] runs, [1] n, [2] last hash, [3] truncated,
  // [4..6] candidates / k-mers / table overflows) with an asynchronous copy in flight to `h_meta`.
  bool pending = false;
  hipEvent_t ev = nullptr;         // recorded behind the sketch's last kernel (on the stream that built it)
  hipStream_t ev_stream = nullptr;
  mg::DevBuf meta;
  uint64_t* h_meta = nullptr;
  int pend_slot = -1;
  uint64_t n_bound = 0;    // upper bound of n while pending (sizes the look-up index)
  uint64_t hmax = 0;
  double expect = 0.0;     // expected candidates: feeds the distinct-count hint at resolution
  struct Redo {            // what rebuilds the sketch on the list path if the counting table overflowed
    const uint8_t* bases = nullptr;
    const uint64_t* offsets = nullptr;
    uint64_t nreads = 0, hmax = 0, s = 0, cap = 0, nbases = 0;
    int k = 0;
    unsigned stage = 0;
    const mg_filter* filter = nullptr;
    // a deferred MERGE (mg_sketch_merge_dev_async) is redone by the sorting merge over its own inputs instead
    bool is_merge = false, use_bound = false;
    const uint64_t* m_hashes = nullptr;
    const uint32_t* m_counts = nullptr;
    uint64_t m_n = 0, m_bound = 0;
  } redo;
  ~mg_sketch();
};

namespace mg {
// ---- the fused multi-k stage-A launch (mg_sketch_multi.hip) ----
struct Slot;          // mg_sketch_dev.h: one 16-byte slot of the counting table
struct MultiKTable {  // one k of the launch: its threshold, counting table (already zeroed), counters and pre-filter
  uint64_t hmax;
  Slot* tab;
  unsigned long long* counters;
  unsigned shift;
  const mg_filter* filter;
  uint32_t epoch = 0;  // != 0: tab is a copy of the filter's resident index and this the epoch of the call
  uint32_t* list = nullptr;  // ... and the list the kernel leaves the touched SLOTS in (filled with 0xffffffff)
  uint64_t listcap = 0;
};
// mg_sketch_cmash.hip: the one-k kernels instantiated for hash definition 1 (mg_set_hash_mode)
int launch_sketch_reads_cmash(int k, unsigned grid, size_t lds, hipStream_t st, const uint8_t* d_bases, const uint64_t* d_offsets,
                              uint64_t nreads, uint64_t hmax, uint64_t* d_cand, uint64_t cap, unsigned long long* d_counters,
                              Slot* d_tab, unsigned bucket_shift, unsigned stage_bytes, const uint32_t* fbits, uint64_t fmask,
                              uint32_t cs_word);
int launch_hash_positions_cmash(int k, unsigned grid, hipStream_t st, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nseq,
                                uint64_t nbases, uint64_t* d_out, bool tagged = false);
int launch_hash_positions_forward(int k, unsigned grid, hipStream_t st, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nseq,
                                  uint64_t nbases, uint64_t* d_out);
bool sketch_reads_multi_supported(const int* ks, int nk);
int launch_sketch_reads_multi(const int* ks, int nk, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t nreads,
                              const MultiKTable* tabs, unsigned stage_bytes);

// Brings a pending sketch's metadata to the host (one stream sync); *rebuilt = 1 when the counting table had
// overflowed and the sketch was recomputed on the list path (anything derived from it while pending is stale).
int sketch_resolve(mg_sketch* sk, int* rebuilt);
// Makes the library's current stream wait (on the device) for the stream that built the sketch, if it is another.
int sketch_wait(const mg_sketch* sk);
}  // namespace mg

// ---- ingest handles (mg_ingest.hip; mg_stream.hip builds them piece by piece) ----
struct mg_reads {
  mg::DevBuf bases, offsets;
  uint64_t nreads = 0, nbases = 0;
};

struct mg_acc_index {
  mg::DevBuf slot_hash, slot_row, names, name_off;
  uint64_t slots = 0;
  uint32_t nacc = 0;
};

#define MG_THIN_MARK 0x01  // first byte of a SEQ field the file reader replaced by its length (never whitespace, never a base)

struct mg_sam_batch {
  mg::DevBuf recs;
  uint64_t nrecs = 0;
  std::string last_qname;
};

namespace mg {
// thin: the piece comes from the file reader's thinning (mg_stream.hip: a SEQ field is MG_THIN_MARK + its length in decimal).
int aln_tokenize_prefix_dev(const uint8_t* d_text, uint64_t nbytes, const mg_acc_index* ix, const char* prev_qname, bool paf,
                            bool final, uint64_t* consumed, mg_sam_batch** out, int* err_kind, uint64_t* err_line, bool thin = false);
}

struct mg_filter {
  // Membership pre-filter over a set of hashes (the genome table's): one bit per hash value modulo the size,
  // nbits = power of two >= 16 x the number of hashes.  No false negatives; ~6 % false positives.
  mg::DevBuf bits;      // u32[nbits / 32]
  uint64_t mask = 0;    // nbits - 1
  unsigned log2_bits = 0;
  // Optional RESIDENT INDEX over the same hashes (mg_filter_make_resident; mg_sketch_dev.h: resident_count): a counting
  // table seeded once with every hash and never cleared, its counters tagged with the epoch of the call that wrote them.
  // A read sketch made with it holds exactly the read k-mers that ARE hashes of the table (the bit filter lets ~6 % of
  // the others through); containment is the same either way.  One copy per stream that sketches with it (the counters of
  // two passes in flight on two streams must not share slots), made on first use.
  struct Resident {
    unsigned shift = 0;
    uint64_t nbuckets = 0, slots = 0, distinct = 0, hmax = 0;
    struct Copy { hipStream_t stream = nullptr; void* slots = nullptr; };
    std::vector<Copy> copies;  // [0]: the seeded one
    ~Resident();
  };
  mutable std::unique_ptr<Resident> resident;
  bool resident_off = false;  // mg_filter_use_resident(f, 0): sketch calls take the bit filter although an index exists
};

struct mg_db {
  // Hash-major layout built once at upload (mg_contain.hip): every (hash, genome) pair of every genome sketch,
  // sorted by hash.  Containment streams the pairs once against the matching run of the read sketch.
  mg::DevBuf pair_hash;  // u64[total]   ascending (equal hashes of different genomes are adjacent)
  mg::DevBuf pair_gen;   // u32[total]   genome of the pair
  mg::DevBuf gsize;      // u32[ngenomes] sketch size of every genome
  mg::DevBuf offsets;    // u64[ngenomes+1]
  uint64_t ngenomes = 0;
  uint64_t total = 0;
  uint64_t max_hash = 0;
};

namespace mg {
// The index over a reference-pipeline table's distinct canonical k_max-mers that stage A BY K-MER IDENTITY reads (mg_kcount.hip,
// mg_kcount_core.h): built once per table on the device from the pairs' k-mers.
struct KmerIndex {
  int k = 0;
  uint64_t ndistinct = 0, nentries = 0, nbuckets = 0;  // (entries: a k-mer filed under several hashes has one for each)
  uint32_t gbits = 0;    // the gate has 2^gbits bits: a hash's bit is number hash >> (32 - gbits)
  uint64_t gate_words = 0;
  uint32_t bmask = 0;    // bucket of a hash = hash & bmask
  DevBuf gate;           // some k-mer of the table is filed under a hash with these leading bits
  DevBuf shared;         // ... and two different such hashes have them (a sample never clears that bit of its copy of the gate)
  DevBuf prim;           // KcEntry[4 x nbuckets]: a bucket's first four entries (mg_kcount_core.h: KcIndexView)
  DevBuf ovf;            // KcEntry[novf + 1]: the later ones
  uint64_t novf = 0;
  DevBuf head;           // u32[npairs]: the pair whose counter holds the occurrences of this pair's k-mer
};
// the main stream is about to read a sample's k-mer counters (mg_kcount.hip): after whoever wrote them last, on whichever stream
int kcounts_wait(const mg_kcounts* kc);
}  // namespace mg

struct mg_refdb {
  // The table of the REFERENCE PIPELINE (mg_refpipe.hip; oracle/mg_oracle.c, "THE REFERENCE'S OWN WIRING"): the hash-major
  // table of the largest k, and for every k below it what derives that k's column from WHICH pairs matched.
  mg_db kmax;
  int nk = 0;
  int ks[4] = {0, 0, 0, 0};
  mg::DevBuf kmer_hi, kmer_lo;  // u64[npairs]: the kept k_max-mers in pair order (a table built here; empty after an upload)
  struct Small {
    mg::DevBuf pa, pb;      // u32[npairs]: number of the k-prefix of the pair's k-mer / of its reverse complement (0xffffffff: none)
    mg::DevBuf cid, cgen;   // u32[ncount]: the distinct (prefix number, genome) combinations, ascending
    mg::DevBuf gsize;       // u32[ngenomes]: distinct prefixes per genome
    uint64_t nprefix = 0, ncount = 0;
    uint64_t marks_at = 0, marks_n = 0;  // this k's bitmap inside `marks`: first word, words
  } small[3];
  mg::DevBuf marks;         // the prefix bitmaps of all smaller k, back to back (zeroed and written by a mark call)
  uint64_t marks_words = 0;
  // the count step's counter copies, zeroed by the mark call that precedes it (one launch less per pass)
  mutable uint32_t* count_part = nullptr;
  mutable uint32_t count_copies = 0;
  mutable uint64_t count_gen = 0;  // the containment call that zeroed them (any later one reuses the buffer)
  // mg_refdb_upload_begin: the arrays still on their way up (settled, and the table checked, by the first call that reads it)
  mutable mg::UploadJob* pending = nullptr;
  mutable bool unchecked = false;
  // (a rank of a multi-GPU job that holds the WHOLE table — stage A by k-mer identity — streams only its share of every count list:
  // entries [n r / W, n (r + 1) / W); the ranks' partial columns are summed by the job's all-reduce.  mg_refdb_set_count_share)
  mutable uint32_t share_rank = 0, share_world = 1;
  mutable int failed = 0;  // the upload or the check failed: every later call on the handle answers this (mg_refpipe.hip: refdb_ready)
  std::unique_ptr<mg::KmerIndex> kidx;  // mg_refdb_index_kmers
  ~mg_refdb();
};
namespace mg { int refdb_ready(const mg_refdb* db); int check_pairs_dev(const mg_db& db); }
