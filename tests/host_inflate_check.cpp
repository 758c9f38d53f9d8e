// The device inflater's decoder (metalign_amd/csrc/mg_inflate_core.h) compiled for the HOST and run lane by lane against zlib:
// whole members as bytes, jobs entered at block starts the finder reports (16-bit symbols resolved with the known window),
// count-only / overflowing jobs, members and trailing garbage, stored and fixed blocks, damaged streams, CRC combination.
// Prints "ok <checks>" or the first failure.  Test infrastructure (tests/test_inflate_core_host.py builds and runs it).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../metalign_amd/csrc/mg_inflate_core.h"

using namespace mgi;

static uint64_t g_rng = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
  g_rng ^= g_rng << 13;
  g_rng ^= g_rng >> 7;
  g_rng ^= g_rng << 17;
  return (uint32_t)(g_rng >> 11);
}

static std::string fastq(size_t nreads) {
  std::string s;
  char buf[64];
  for (size_t i = 0; i < nreads; ++i) {
    snprintf(buf, sizeof(buf), "@read%zu/1 len=150\n", i);
    s += buf;
    for (int j = 0; j < 150; ++j) s += "ACGT"[rnd() & 3];
    s += "\n+\n";
    for (int j = 0; j < 150; ++j) s += (char)(35 + (rnd() % 7 == 0 ? rnd() % 39 : 37));
    s += "\n";
  }
  return s;
}

static std::string deflate_raw(const std::string& in, int level, int strategy = Z_DEFAULT_STRATEGY, int wbits = 15 + 16, int memlevel = 8) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, level, Z_DEFLATED, wbits, memlevel, strategy) != Z_OK) abort();
  std::string out(deflateBound(&zs, in.size()) + 64, '\0');
  zs.next_in = (Bytef*)in.data();
  zs.avail_in = (uInt)in.size();
  zs.next_out = (Bytef*)out.data();
  zs.avail_out = (uInt)out.size();
  if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
  out.resize(zs.total_out);
  deflateEnd(&zs);
  return out;
}

struct Words {
  std::vector<uint32_t> w;
  uint64_t nbytes;
  explicit Words(const std::string& s) : w((s.size() + 3) / 4 + 1, 0u), nbytes(s.size()) { memcpy(w.data(), s.data(), s.size()); }
};

static int g_checks = 0;
#define CHECK(cond, ...)                                   \
  do {                                                     \
    ++g_checks;                                            \
    if (!(cond)) {                                         \
      printf("FAIL %s:%d: ", __FILE__, __LINE__);          \
      printf(__VA_ARGS__);                                 \
      printf("\n");                                        \
      exit(1);                                             \
    }                                                      \
  } while (0)

static Shared g_sh;

// the whole stream as one job of bytes
static Result whole(const std::string& gz, std::string* out, std::vector<Event>* evs, uint32_t flags = F_HEADER | F_MEMBER_START,
                    uint64_t cap = ~0ull, bool final_input = true) {
  Words in(gz);
  HostExec ex;
  ex.sh = &g_sh;
  Job job{0, ~0ull, 0, 0, flags, 0};
  std::vector<uint8_t> buf(cap == ~0ull ? (size_t)8 << 20 : (size_t)cap + 1);
  job.out_cap = cap == ~0ull ? buf.size() : cap;
  Result res;
  memset(&res, 0, sizeof(res));
  std::vector<Event> events(64);
  uint32_t nev = 0;
  run_job<HostExec, uint8_t>(ex, g_sh, in.w.data(), in.nbytes, final_input, job, 0, buf.data(), &res, events.data(), &nev, (uint32_t)events.size());
  {  // the same job by the one-job-per-lane decoder: the same result, the same bytes, the same member ends
    static LaneMem lmem;
    static LaneScratch lscr;
    std::vector<uint8_t> buf2(buf.size());
    std::vector<Event> ev2(64);
    uint32_t nev2 = 0;
    LaneDec<uint8_t> d;
    d.init(in.w.data(), in.nbytes, final_input, job, 0, buf2.data(), &lmem, &lscr, ev2.data(), &nev2, (uint32_t)ev2.size());
    while (d.state != LS_DONE) d.round();
    Result r2;
    d.result(&r2);
    CHECK(r2.status == res.status && (res.status >= ST_ERR || (r2.out_count == res.out_count && r2.overflow == res.overflow && r2.end_bit == res.end_bit &&
                                                                   r2.crc == res.crc && r2.isize == res.isize && nev2 == nev)),
          "the lane decoder: status %u/%u count %llu/%llu overflow %u/%u end %llu/%llu events %u/%u", r2.status, res.status, (unsigned long long)r2.out_count,
          (unsigned long long)res.out_count, r2.overflow, res.overflow, (unsigned long long)r2.end_bit, (unsigned long long)res.end_bit, nev2, nev);
    if (!res.overflow && !(flags & F_COUNT_ONLY) && res.status < ST_ERR) CHECK(memcmp(buf.data(), buf2.data(), (size_t)res.out_count) == 0, "the lane decoder: bytes");
    for (uint32_t i = 0; res.status < ST_ERR && i < nev && i < 64; ++i)
      CHECK(ev2[i].out_pos == events[i].out_pos && ev2[i].crc == events[i].crc && ev2[i].isize == events[i].isize, "the lane decoder: member end %u", i);
  }
  if (out) out->assign((const char*)buf.data(), (size_t)(res.overflow ? 0 : res.out_count));
  if (evs) evs->assign(events.begin(), events.begin() + (nev < events.size() ? nev : events.size()));
  return res;
}

static void check_whole(const char* what, const std::string& text, const std::string& gz) {
  std::string got;
  std::vector<Event> ev;
  const Result r = whole(gz, &got, &ev);
  CHECK(r.status == ST_END, "%s: status %u", what, r.status);
  CHECK(got.size() == text.size() && got == text, "%s: %zu bytes against %zu", what, got.size(), text.size());
  CHECK(ev.size() == 1 && ev[0].crc == (uint32_t)crc32(0, (const Bytef*)text.data(), (uInt)text.size()) && ev[0].isize == (uint32_t)text.size() &&
            ev[0].out_pos == text.size(), "%s: trailer", what);
  CHECK(r.end_bit == gz.size() * 8, "%s: end_bit", what);
  // count only, and a capacity that runs out: the same count, nothing written beyond the capacity
  const Result c = whole(gz, nullptr, nullptr, F_HEADER | F_MEMBER_START | F_COUNT_ONLY);
  CHECK(c.status == ST_END && c.out_count == text.size() && !c.overflow, "%s: count-only", what);
  if (text.size() > 10) {
    const Result o = whole(gz, nullptr, nullptr, F_HEADER | F_MEMBER_START, text.size() / 2);
    CHECK(o.status == ST_END && o.out_count == text.size() && o.overflow == 1, "%s: overflow", what);
  }
}

// jobs entered at every block start the finder reports in [from, to): 16-bit symbols, resolved with the true window
static int check_entered(const char* what, const std::string& text, const std::string& gz, uint64_t step_bytes) {
  Words in(gz);
  HostExec ex;
  ex.sh = &g_sh;
  // the true chain of block boundaries, from a byte-mode pass that stops at every block
  std::vector<uint64_t> bounds, outs;
  {
    uint64_t at = 0, outpos = 0;
    std::vector<uint8_t> buf(text.size() + 1);
    uint32_t flags = F_HEADER | F_MEMBER_START;
    for (;;) {
      Job job{at, at + 1, 0, buf.size(), flags, 0};
      Result res;
      std::vector<Event> events(8);
      uint32_t nev = 0;
      // (a job that starts inside the member cannot run in byte mode: count only)
      job.flags |= at ? F_COUNT_ONLY : 0;
      run_job<HostExec, uint8_t>(ex, g_sh, in.w.data(), in.nbytes, true, job, 0, buf.data(), &res, events.data(), &nev, 8);
      if (res.status != ST_STOP) break;
      outpos += res.out_count;
      bounds.push_back(res.end_bit);
      outs.push_back(outpos);
      at = res.end_bit;
      flags = 0;
    }
  }
  int entered = 0;
  std::vector<uint16_t> sym(text.size() + 1);
  for (uint64_t from = step_bytes * 8; from + 64 < gz.size() * 8; from += step_bytes * 8) {
    // first plausible block start at or behind `from`, inside the next step_bytes
    uint64_t found = ~0ull;
    for (uint64_t p = from; p < from + step_bytes * 8 && p + 64 < gz.size() * 8; ++p) {
      if ((p & 31u) == 0) {  // the finder's first test, 32 positions per word operation, against the same test position by position
        const uint64_t w = p >> 5;
        const uint64_t x = (uint64_t)in.w[w] | (w + 1 < in.w.size() ? (uint64_t)in.w[w + 1] << 32 : 0ull);
        const uint32_t mask = probe_fields_mask(x);
        for (uint32_t i = 0; i < 32; ++i) CHECK(((mask >> i) & 1u) == (probe_fields(x >> i) ? 1u : 0u), "%s: probe_fields_mask differs at bit %llu", what, (unsigned long long)(p + i));
      }
      if (probe_block_start(in.w.data(), in.w.size(), p, false)) {  // by the format's rules alone (the finder's second try)
        const bool loose = validate_block_start(ex, g_sh, in.w.data(), in.nbytes, p, false);
        CHECK(light_validate(in.w.data(), in.nbytes, p, false) == loose, "%s: the two validators disagree (loose) at bit %llu (%d)", what, (unsigned long long)p, (int)loose);
        CHECK(loose || !validate_block_start(ex, g_sh, in.w.data(), in.nbytes, p), "%s: strict accepts what loose refuses at bit %llu", what, (unsigned long long)p);
      } else {
        CHECK(!probe_block_start(in.w.data(), in.w.size(), p), "%s: the strict probe accepts what the loose one refuses at bit %llu", what, (unsigned long long)p);
      }
      if (!probe_block_start(in.w.data(), in.w.size(), p)) continue;
      const bool heavy = validate_block_start(ex, g_sh, in.w.data(), in.nbytes, p);
      CHECK(light_validate(in.w.data(), in.nbytes, p) == heavy, "%s: the two validators disagree at bit %llu (%d)", what, (unsigned long long)p, (int)heavy);
      if (!heavy) continue;
      found = p;
      break;
    }
    // every true boundary of a non-final dynamic block in the range must have been found no later than it stands
    size_t bi = 0;
    while (bi < bounds.size() && bounds[bi] < from) ++bi;
    if (found == ~0ull) continue;
    CHECK(bi < bounds.size() && bounds[bi] >= found, "%s: the finder skipped a boundary (%llu vs %llu)", what,
          (unsigned long long)(bi < bounds.size() ? bounds[bi] : 0), (unsigned long long)found);
    if (bounds[bi] != found) continue;  // a false positive in front of the true boundary: the chain check of the caller's business
    const uint64_t outpos = outs[bi];
    Job job{found, ~0ull, 0, sym.size(), 0, 0};
    Result res;
    std::vector<Event> events(8);
    uint32_t nev = 0;
    alignas(16) static uint16_t tail[32768];
    run_job<HostExec, uint16_t>(ex, g_sh, in.w.data(), in.nbytes, true, job, 0, sym.data(), &res, events.data(), &nev, 8, tail);
    CHECK(res.status == ST_END, "%s: entered job status %u", what, res.status);
    {  // ... and by the lane decoder
      static LaneMem lmem;
      static LaneScratch lscr;
      std::vector<uint16_t> sym2(sym.size());
      LaneDec<uint16_t> d;
      uint32_t nev2 = 0;
      d.init(in.w.data(), in.nbytes, true, job, 0, sym2.data(), &lmem, &lscr, events.data(), &nev2, 8);
      while (d.state != LS_DONE) d.round();
      CHECK(d.status == res.status && d.outn == res.out_count && memcmp(sym.data(), sym2.data(), (size_t)res.out_count * 2) == 0, "%s: the lane decoder's symbols", what);
    }
    for (uint32_t w = 0; w < 32768; ++w) {  // the window behind the job, as the job knows it
      const uint64_t n = res.out_count;
      const uint16_t want = n >= 32768 ? sym[n - 32768 + w] : (w < 32768 - n ? (uint16_t)(0x8000 | (w + n)) : sym[w - (32768 - n)]);
      if (tail[w] != want) CHECK(false, "%s: tail symbol %u", what, w);
    }
    CHECK(outpos + res.out_count == text.size(), "%s: entered job count %llu + %llu vs %zu", what, (unsigned long long)outpos,
          (unsigned long long)res.out_count, text.size());
    for (uint64_t i = 0; i < res.out_count; ++i) {
      const uint16_t v = sym[i];
      uint8_t b;
      if (v < 256) b = (uint8_t)v;
      else {
        CHECK(v >= 0x8000, "%s: symbol %x", what, v);
        const int64_t q = (int64_t)outpos - 32768 + (v & 0x7fff);
        CHECK(q >= 0 && q < (int64_t)outpos, "%s: window symbol out of range", what);
        b = (uint8_t)text[(size_t)q];
      }
      if (b != (uint8_t)text[outpos + i]) CHECK(false, "%s: entered at bit %llu: byte %llu differs", what, (unsigned long long)found, (unsigned long long)i);
    }
    ++g_checks;
    ++entered;
  }
  return entered;
}

// the 16-bit table entries of the window decode (lit16 / dist16) say what the 32-bit entries of the scalar path say, for every symbol and
// every code length a table slot can hold; what is not for the window decode (end of block, a symbol that does not exist) says so
static void check_entries() {
  for (uint32_t l = 1; l <= (uint32_t)LB; ++l)
    for (uint32_t s = 0; s < 288; ++s) {
      const uint32_t w = wide_lit(lit16(s, l)), e = lit_entry(s, l);
      if (s == 256) { CHECK(w == (l | (K_EOB << 8)), "entries: end of block, length %u: %x", l, w); continue; }
      if (s > 285) { CHECK(w == 0u, "entries: literal/length symbol %u: %x", s, w); continue; }
      CHECK(w == e, "entries: literal/length symbol %u, length %u: %x vs %x", s, l, w, e);
      CHECK(lit16(s, l) <= 0xffffu && (lit16(s, l) & 15u) == l, "entries: literal/length symbol %u does not fit", s);
    }
  for (uint32_t l = 1; l <= (uint32_t)DB; ++l)
    for (uint32_t s = 0; s < 32; ++s) {
      const uint32_t w = wide_dist(dist16(s, l)), e = dist_entry(s, l);
      if (s > 29) { CHECK(w == 0u, "entries: distance symbol %u: %x", s, w); continue; }
      CHECK(w == e && dist16(s, l) <= 0xffffu, "entries: distance symbol %u, length %u: %x vs %x", s, l, w, e);
    }
  CHECK(wide_lit(K_LONG << 4) == (K_LONG << 8 | 15u) && wide_dist(K_LONG << 4) == (K_LONG << 8 | 15u) && wide_lit(0) == 0u && wide_dist(0) == 0u,
        "entries: %s", "long / empty");
}

int main() {
  check_entries();
  const std::string fq = fastq(6000);  // ~1.9 MB
  std::string bin(300000, '\0');
  for (auto& c : bin) c = (char)rnd();
  std::string zeros(500000, '\0');
  std::string runs;
  for (int i = 0; i < 40000; ++i) runs += std::string(1 + rnd() % 300, (char)('A' + rnd() % 4));
  std::string few = "ab";
  for (int i = 0; i < 12; ++i) few += few;  // two symbols only
  const std::string mixed = fq.substr(0, 400000) + bin.substr(0, 100000) + zeros.substr(0, 70000) + fq.substr(400000, 300000);

  struct Case { const char* name; const std::string* text; };
  const Case cases[] = {{"fastq", &fq}, {"binary", &bin}, {"zeros", &zeros}, {"runs", &runs}, {"few", &few}, {"mixed", &mixed}};
  for (const Case& c : cases) {
    for (int level : {0, 1, 4, 6, 9}) {
      char what[64];
      snprintf(what, sizeof(what), "%s level %d", c.name, level);
      check_whole(what, *c.text, deflate_raw(*c.text, level));
    }
    char what[64];
    snprintf(what, sizeof(what), "%s fixed", c.name);
    check_whole(what, *c.text, deflate_raw(*c.text, 6, Z_FIXED));
    snprintf(what, sizeof(what), "%s huffman-only", c.name);
    check_whole(what, *c.text, deflate_raw(*c.text, 6, Z_HUFFMAN_ONLY));
    snprintf(what, sizeof(what), "%s rle", c.name);
    check_whole(what, *c.text, deflate_raw(*c.text, 6, Z_RLE));
    snprintf(what, sizeof(what), "%s small blocks", c.name);
    check_whole(what, *c.text, deflate_raw(*c.text, 6, Z_DEFAULT_STRATEGY, 15 + 16, 1));  // memLevel 1: blocks of 128 symbols... many headers
  }
  for (size_t n : {0ul, 1ul, 2ul, 3ul, 63ul, 64ul, 65ul, 257ul, 258ul, 259ul, 4095ul, 4096ul, 4097ul, 32768ul, 32769ul, 65536ul}) {
    const std::string t = fq.substr(0, n);
    char what[64];
    snprintf(what, sizeof(what), "prefix %zu", n);
    check_whole(what, t, deflate_raw(t, 6));
    const std::string z = zeros.substr(0, n);
    snprintf(what, sizeof(what), "zeros %zu", n);
    check_whole(what, z, deflate_raw(z, 9));
  }
  // jobs entered in the middle
  int entered = 0;
  entered += check_entered("fastq 6", fq, deflate_raw(fq, 6), 20000);
  entered += check_entered("fastq 1", fq, deflate_raw(fq, 1), 30000);
  entered += check_entered("fastq 9", fq, deflate_raw(fq, 9), 20000);
  entered += check_entered("mixed 6", mixed, deflate_raw(mixed, 6), 15000);
  entered += check_entered("runs 6", runs, deflate_raw(runs, 6), 3000);
  CHECK(entered > 20, "only %d jobs entered in the middle", entered);

  // members, padding, garbage, header fields
  {
    const std::string a = fq.substr(0, 500000), b = fq.substr(500000, 100), c = fq.substr(500100, 300000);
    const std::string gz = deflate_raw(a, 6) + deflate_raw(b, 1) + deflate_raw("", 6) + deflate_raw(c, 4);
    std::string got;
    std::vector<Event> ev;
    Result r = whole(gz, &got, &ev);
    CHECK(r.status == ST_END && got == a + b + c && ev.size() == 4, "members: status %u, %zu events", r.status, ev.size());
    CHECK(ev[0].out_pos == a.size() && ev[1].out_pos == a.size() + b.size() && ev[2].out_pos == ev[1].out_pos && ev[3].out_pos == got.size(), "member ends");
    CHECK(ev[1].crc == (uint32_t)crc32(0, (const Bytef*)b.data(), (uInt)b.size()) && ev[2].isize == 0, "member trailers");
    for (const std::string& pad : {std::string(1, '\0'), std::string(9, '\0'), std::string(4000, '\0'), std::string("trailing garbage"), std::string("\x1f"), std::string("\n")}) {
      r = whole(gz + pad, &got, &ev);
      if (pad == "\x1f") { CHECK(r.status == ST_TRUNC, "a cut magic number: status %u", r.status); continue; }
      CHECK(r.status == ST_END && got == a + b + c, "padding of %zu: status %u", pad.size(), r.status);
    }
    // the same bytes, not final: more may come
    r = whole(gz, &got, &ev, F_HEADER | F_MEMBER_START, ~0ull, false);
    CHECK(r.status == ST_NEED_MORE && r.out_count == got.size(), "not final: status %u", r.status);
    // one member only
    r = whole(gz, &got, &ev, F_HEADER | F_MEMBER_START | F_ONE_MEMBER);
    CHECK(r.status == ST_MEMBER && got == a && r.isize == a.size() && r.crc == (uint32_t)crc32(0, (const Bytef*)a.data(), (uInt)a.size()), "one member");
    // header fields
    const std::string raw = deflate_raw(a, 6, Z_DEFAULT_STRATEGY, -15);
    std::string hdr("\x1f\x8b\x08", 3);
    hdr += (char)(2 | 4 | 8 | 16);
    hdr += std::string("\0\0\0\0\0\x03", 6) + std::string("\x05\0hello", 7) + std::string("name.fq\0", 8) + std::string("a comment\0", 10) + std::string("\x12\x34", 2);
    std::string trailer(8, '\0');
    const uint32_t cr = (uint32_t)crc32(0, (const Bytef*)a.data(), (uInt)a.size()), sz = (uint32_t)a.size();
    memcpy(&trailer[0], &cr, 4);
    memcpy(&trailer[4], &sz, 4);
    r = whole(hdr + raw + trailer, &got, &ev);
    CHECK(r.status == ST_END && got == a, "header fields: status %u", r.status);
  }
  // damaged streams: an error status, or (a flipped literal) a different CRC — never a crash, never a hang
  {
    const std::string t = fq.substr(0, 300000);
    const std::string gz = deflate_raw(t, 6);
    std::string got;
    std::vector<Event> ev;
    for (size_t cutat : {gz.size() - 1, gz.size() - 8, gz.size() - 9, gz.size() / 2, (size_t)12, (size_t)6, (size_t)2, (size_t)1}) {
      const Result r = whole(gz.substr(0, cutat), &got, &ev);
      CHECK(r.status == ST_TRUNC, "cut at %zu: status %u", cutat, r.status);
    }
    CHECK(whole("plain text, not gzip", &got, &ev).status == ST_BAD_HEADER, "not gzip");
    CHECK(whole("", &got, &ev).status == ST_BAD_HEADER, "empty input");
    int caught = 0, crc_only = 0;
    for (int k = 0; k < 200; ++k) {
      std::string bad = gz;
      bad[20 + rnd() % (bad.size() - 40)] ^= (char)(1u << (rnd() & 7));
      const Result r = whole(bad, &got, &ev);
      if (r.status >= ST_ERR) ++caught;
      else if (r.status == ST_END && (ev.empty() || ev[0].crc != (uint32_t)crc32(0, (const Bytef*)got.data(), (uInt)got.size()) || ev[0].isize != got.size())) ++crc_only;
      else CHECK(false, "a flipped bit went unnoticed (status %u)", r.status);
    }
    CHECK(caught + crc_only == 200 && caught > 0, "flips");
  }
  // CRC combination
  {
    uint32_t x2n[32];
    crc_make_x2n(x2n);
    for (size_t cutat : {(size_t)0, (size_t)1, (size_t)4096, fq.size() / 3, fq.size() - 1, fq.size()}) {
      const uint32_t c1 = (uint32_t)crc32(0, (const Bytef*)fq.data(), (uInt)cutat);
      const uint32_t c2 = (uint32_t)crc32(0, (const Bytef*)fq.data() + cutat, (uInt)(fq.size() - cutat));
      CHECK(crc_combine(c1, c2, fq.size() - cutat, x2n) == (uint32_t)crc32(0, (const Bytef*)fq.data(), (uInt)fq.size()), "crc combine at %zu", cutat);
    }
  }
  printf("ok %d checks, %d jobs entered in the middle\n", g_checks, entered);
  return 0;
}
