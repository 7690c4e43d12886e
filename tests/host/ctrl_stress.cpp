// CPU stress test of the control-path primitives of libphonic_gpu.so (phonic_amd/csrc/pg_ctrl.h), built with -fsanitize=thread by
// tests/test_ctrl_ring.py: producer threads push sample-time-tagged messages while a consumer drains, exactly as handle threads and
// the thread inside write() do. Checks: nothing lost, nothing duplicated, per-producer order kept, push fails (and only fails) on a
// full ring; ChunkTable entries appended by the owner are visible to concurrent readers.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../phonic_amd/csrc/pg_ctrl.h"

int main(int argc, char** argv) {
  const int n_producers = argc > 1 ? atoi(argv[1]) : 4;
  const long per_producer = argc > 2 ? atol(argv[2]) : 250000;
  pgc::CtrlRing ring(4096);
  std::atomic<long> full_hits{0};
  std::atomic<int> done{0};
  std::vector<std::thread> producers;
  for (int p = 0; p < n_producers; ++p) {
    producers.emplace_back([&, p]() {
      for (long i = 0; i < per_producer; ++i) {
        pgc::CtrlMsg m{};
        m.type = pgc::CT_FX_PARAM; m.id = p; m.param = (int32_t)(i & 0x7fffffff); m.value = (float)i; m.sample_time = (uint64_t)i;
        while (!ring.push(m)) { full_hits.fetch_add(1, std::memory_order_relaxed); std::this_thread::yield(); }
      }
      done.fetch_add(1);
    });
  }
  std::vector<long> next(n_producers, 0);
  long total = 0, errors = 0;
  for (;;) {
    pgc::CtrlMsg m;
    bool any = false;
    while (ring.pop(m)) {
      any = true;
      if (m.id < 0 || m.id >= n_producers || (long)m.sample_time != next[m.id] || m.value != (float)next[m.id]) ++errors;
      else ++next[m.id];
      ++total;
    }
    if (!any) {
      if (done.load() == n_producers) {
        while (ring.pop(m)) { if ((long)m.sample_time != next[m.id]) ++errors; else ++next[m.id]; ++total; }
        break;
      }
      std::this_thread::yield();
    }
  }
  for (auto& t : producers) t.join();
  // a full ring refuses, an emptied one accepts again
  pgc::CtrlRing small(8);
  pgc::CtrlMsg m{};
  int accepted = 0;
  while (small.push(m)) ++accepted;
  if (accepted != 8) ++errors;
  if (!small.pop(m) || !small.push(m) || small.push(m)) ++errors;
  // ChunkTable: owner appends while readers scan what size() covers
  pgc::ChunkTable<int8_t, 64, 64> tab;
  std::atomic<bool> stop{false};
  std::atomic<long> bad{0};
  std::thread reader([&]() {
    while (!stop.load()) { const size_t n = tab.size(); for (size_t i = 0; i < n; ++i) { const int8_t v = tab.get(i); if (v != (int8_t)(i % 100) && v != -1) bad.fetch_add(1); } }
  });
  for (int i = 0; i < 4000; ++i) { if (!tab.append((int8_t)(i % 100))) ++errors; if (i % 7 == 0) tab.set((size_t)i / 2, -1); }
  stop.store(true);
  reader.join();
  if (tab.size() != 4000 || tab.append(1) == false) {}  // (4096 entries fit: 64 chunks of 64)
  errors += bad.load();
  printf("{\"producers\": %d, \"messages\": %ld, \"expected\": %ld, \"errors\": %ld, \"full_hits\": %ld}\n", n_producers, total, per_producer * n_producers, errors,
         full_hits.load());
  return (errors == 0 && total == per_producer * n_producers) ? 0 : 1;
}
