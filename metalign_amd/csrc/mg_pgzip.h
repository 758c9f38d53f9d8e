// mg_pgzip.h — one gzip stream inflated by many host threads (mg_pgzip.hip).  Not part of the ABI.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mg {

class PGzip {
 public:
  struct Chunk;
  struct Pools;
  struct Queue;
  // fd: the open gzip file of fsize bytes (closed by the object when own_fd); nthreads inflating threads; chunk_bytes of
  // COMPRESSED data per unit of work.  nullptr + *err on failure.  Decoding starts at once, in the background.
  static std::unique_ptr<PGzip> open(int fd, bool own_fd, uint64_t fsize, int nthreads, uint64_t chunk_bytes, std::string* err);
  ~PGzip();
  // The next bytes of the inflated stream (every member; trailing garbage after the last member ignored): up to cap of
  // them, fewer only at the end of the stream (0 = nothing left); -1 on error (error()).  One consumer thread.
  int64_t read(uint8_t* dst, uint64_t cap);
  std::string error();

 private:
  PGzip();
  void run();
  void fail_with(const std::string& e);
  int fd_ = -1;
  bool own_fd_ = false;
  const uint8_t* map_ = nullptr;
  uint64_t size_ = 0, chunk_ = 2u << 20;
  int nthreads_ = 1;
  std::thread coordinator_;
  std::mutex m_;
  std::condition_variable cv_data_, cv_space_;
  std::unique_ptr<Queue> queue_;             // inflated bytes in stream order
  uint64_t front_off_ = 0, queued_bytes_ = 0;
  uint64_t max_queued_ = 1ull << 30;         // the coordinator starts no new batch while this much waits to be read
  bool done_ = false, stop_ = false;
  std::string error_;
  std::unique_ptr<Pools> pools_;
};

}  // namespace mg
