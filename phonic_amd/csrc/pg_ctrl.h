// Control path of the mixer graph: what may be called from ANY thread while another thread is inside write().
//
// The reference's handles (EffectHandle::set_parameter, FilePlaybackHandle::set_volume / set_panning / set_speed / seek / stop,
// src/player/handles/{effect,file}.rs) do nothing but push a MixerMessage into the mixer's lock-free
// `ArrayQueue<MixerMessage>` (capacity 4096, src/source/mixed.rs:233-234); the audio thread drains the queue at the top of
// MixedSource::write (process_messages, mixed.rs:294-499) and turns sample-time-tagged messages into sorted events. This header is the
// same split for the C ABI: a bounded multi-producer / single-consumer ring of plain records (`CtrlRing`, after D. Vyukov's bounded
// queue: one sequence word per cell, no locks, no allocation after construction) plus append-only id tables that producer threads may
// read while the owner thread adds effects and voices (`ChunkTable`). Plain C++17, no HIP: tests/host/ctrl_stress.cpp builds it with
// -fsanitize=thread on the CPU.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <memory>

namespace pgc {

enum CtrlType : int32_t {
  CT_FX_PARAM = 0,      // id = effect id, param = parameter index, value = resolved raw value   (ProcessEffectParameterUpdate)
  CT_FX_RESET = 1,      // id = effect id                                                        (ProcessEffectMessage: Reset)
  CT_VOICE_VOLUME = 2,  // id = voice id, value                                                  (SetSourceVolume)
  CT_VOICE_PAN = 3,     //                                                                       (SetSourcePanning)
  CT_VOICE_SPEED = 4,   // dvalue = speed, value = glide (semitones / s, <= 0: none)             (SetSourceSpeed)
  CT_VOICE_SEEK = 5,    // dvalue = position in seconds                                          (SeekSource)
  CT_VOICE_STOP = 6,    // sample_time = stop time (0: now); a message, not an event             (StopSource)
  CT_STOP_ALL = 7,      // Player::stop_all_sources: stop every source + RemoveAllPendingEvents
  CT_VOICE_REMOVE = 8,  // id = voice id                                                         (RemoveSource)
};

struct CtrlMsg {
  int32_t type, id, param, pad;
  float value, value2;
  double dvalue;
  uint64_t sample_time;
};

// Bounded lock-free queue: any number of producers, ONE consumer. push() fails when `capacity` messages are waiting (the reference
// returns Error::SendError from the handle in that case). A producer that was pre-empted between claiming a cell and publishing it
// makes pop() report "empty" for the cells behind it until it publishes: messages are never lost or reordered per producer.
class CtrlRing {
 public:
  explicit CtrlRing(size_t capacity_pow2) : mask_(capacity_pow2 - 1), cells_(new Cell[capacity_pow2]) {
    for (size_t i = 0; i <= mask_; ++i) cells_[i].seq.store(i, std::memory_order_relaxed);
    enqueue_.store(0, std::memory_order_relaxed);
    dequeue_ = 0;
  }
  size_t capacity() const { return mask_ + 1; }
  bool push(const CtrlMsg& m) {
    uint64_t pos = enqueue_.load(std::memory_order_relaxed);
    for (;;) {
      Cell& c = cells_[pos & mask_];
      const uint64_t seq = c.seq.load(std::memory_order_acquire);
      const int64_t dif = (int64_t)seq - (int64_t)pos;
      if (dif == 0) {
        if (enqueue_.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
          c.msg = m;
          c.seq.store(pos + 1, std::memory_order_release);
          return true;
        }
      } else if (dif < 0) {
        return false;  // full
      } else {
        pos = enqueue_.load(std::memory_order_relaxed);
      }
    }
  }
  bool pop(CtrlMsg& out) {  // consumer thread only
    Cell& c = cells_[dequeue_ & mask_];
    const uint64_t seq = c.seq.load(std::memory_order_acquire);
    if ((int64_t)seq - (int64_t)(dequeue_ + 1) != 0) return false;  // empty (or the next producer has not published yet)
    out = c.msg;
    c.seq.store(dequeue_ + mask_ + 1, std::memory_order_release);
    ++dequeue_;
    return true;
  }

 private:
  struct Cell {
    std::atomic<uint64_t> seq;
    CtrlMsg msg;
  };
  const size_t mask_;
  std::unique_ptr<Cell[]> cells_;
  alignas(64) std::atomic<uint64_t> enqueue_;
  alignas(64) uint64_t dequeue_;
};

// Append-only table of small atomics indexed by id. The owner thread appends (add_effect / add_voice) and may overwrite entries
// (removal); any thread may read entries below size() at any time: chunks are never moved or freed before the table dies.
template <class T, size_t CHUNK = 4096, size_t MAX_CHUNKS = 4096>
class ChunkTable {
 public:
  ChunkTable() {
    for (size_t i = 0; i < MAX_CHUNKS; ++i) chunks_[i].store(nullptr, std::memory_order_relaxed);
    size_.store(0, std::memory_order_relaxed);
  }
  ~ChunkTable() {
    for (size_t i = 0; i < MAX_CHUNKS; ++i) delete[] chunks_[i].load(std::memory_order_relaxed);
  }
  ChunkTable(const ChunkTable&) = delete;
  ChunkTable& operator=(const ChunkTable&) = delete;
  size_t size() const { return size_.load(std::memory_order_acquire); }
  bool append(T v) {  // owner thread only
    const size_t i = size_.load(std::memory_order_relaxed);
    if (i / CHUNK >= MAX_CHUNKS) return false;
    std::atomic<T>* c = chunks_[i / CHUNK].load(std::memory_order_relaxed);
    if (!c) {
      c = new std::atomic<T>[CHUNK];
      for (size_t k = 0; k < CHUNK; ++k) c[k].store(T(), std::memory_order_relaxed);
      chunks_[i / CHUNK].store(c, std::memory_order_release);
    }
    c[i % CHUNK].store(v, std::memory_order_relaxed);
    size_.store(i + 1, std::memory_order_release);
    return true;
  }
  void set(size_t i, T v) { chunks_[i / CHUNK].load(std::memory_order_acquire)[i % CHUNK].store(v, std::memory_order_release); }
  T get(size_t i) const { return chunks_[i / CHUNK].load(std::memory_order_acquire)[i % CHUNK].load(std::memory_order_acquire); }

 private:
  std::atomic<std::atomic<T>*> chunks_[MAX_CHUNKS];
  std::atomic<size_t> size_;
};

}  // namespace pgc
