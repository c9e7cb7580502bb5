// CPU-only self-test harness for host logic that has no C-ABI entry of its own (run by tests/test_host_cpp.py, no GPU needed):
//   host_selftest mergequeue <threads> <rounds>     stress of merge_queue.h: every request served exactly once, merged batches
//   host_selftest tokenize <tokens.json> [manifest]  stdin lines -> "n | w0 w1 ... | id0 id1 ..." (PuncTokenizerHip::Tokenize)
//   host_selftest jsonstrings <file>                 prints the strings of the first JSON array, one per line, as hex bytes
//   host_selftest hotwords <tokens.json> <seg_dict|->   stdin lines of hotword strings -> "n | len ... | id ..." (HotwordIdMatrix)
//   host_selftest vocabtext <tokens.json> [language]  stdin lines of ids -> HostVocab::Vector2StringV2 of each line (ONE object:
//                                                     the end-of-call memory carries over), as hex bytes
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../merge_queue.h"
#include "ct_transformer_hip.h"
#include "host_vocab.h"
#include "hotword_text.h"
#include "json_strings.h"

namespace {

struct Req : pfhip_detail::MergeReqBase {
  int thread = 0, value = 0, result = -1, batch = 0;
};

int run_mergequeue(int n_threads, int rounds) {
  pfhip_detail::MergeQueue<Req> q;
  std::atomic<long> execs{0}, served{0}, max_batch{0};
  std::atomic<int> errors{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t)
    pool.emplace_back([&, t] {
      for (int r = 0; r < rounds; ++r) {
        Req me;
        me.thread = t; me.value = t * 1000 + r;
        q.submit(
            me, 500, [&](const std::deque<Req*>& dq) { return (int)dq.size() >= n_threads; },
            [&](std::deque<Req*>& dq, std::vector<Req*>& take) {
              std::deque<Req*> later;
              while (!dq.empty() && (int)take.size() < 48) {          // cap below the thread count: leftovers must be promoted
                Req* x = dq.front(); dq.pop_front();
                bool dup = false;
                for (Req* y : take) dup = dup || y->thread == x->thread;
                if (dup) later.push_back(x); else take.push_back(x);
              }
              for (auto it = later.rbegin(); it != later.rend(); ++it) dq.push_front(*it);
            },
            [&](std::vector<Req*>& take) {
              ++execs;
              long mb = max_batch.load();
              while ((long)take.size() > mb && !max_batch.compare_exchange_weak(mb, (long)take.size())) {}
              for (Req* x : take) {
                if (x->result != -1) ++errors;                        // served twice
                x->result = 2 * x->value + 1;
                x->batch = (int)take.size();
              }
            });
        if (me.result != 2 * me.value + 1) ++errors;
        ++served;
      }
    });
  for (auto& th : pool) th.join();
  std::printf("served %ld execs %ld max_batch %ld errors %d leftover %zu leader_active %d\n", served.load(), execs.load(), max_batch.load(),
              errors.load(), q.q.size(), (int)q.leader_active);
  return errors.load() == 0 && served.load() == (long)n_threads * rounds && q.q.empty() && !q.leader_active ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  if (cmd == "mergequeue") return run_mergequeue(argc > 2 ? std::atoi(argv[2]) : 64, argc > 3 ? std::atoi(argv[3]) : 200);
  if (cmd == "tokenize" && argc >= 3) {
    funasr::PuncTokenizerHip tk;
    std::string man = "{}";
    if (argc > 3) { std::ifstream f(argv[3]); std::stringstream ss; ss << f.rdbuf(); man = ss.str(); }
    if (!tk.Open(man, argv[2])) { std::fprintf(stderr, "cannot open tokens\n"); return 1; }
    std::string line;
    while (std::getline(std::cin, line)) {
      std::vector<std::string> words;
      std::vector<int> ids;
      tk.Tokenize(line.c_str(), words, ids);
      std::printf("%zu |", words.size());
      for (const std::string& w : words) std::printf(" %s", w.c_str());
      std::printf(" |");
      for (int id : ids) std::printf(" %d", id);
      std::printf("\n");
    }
    std::printf("punc %s %s %s %s %s %s ispunc %d %d\n", tk.Id2Punc(0).c_str(), tk.Id2Punc(1).c_str(), tk.Id2Punc(2).c_str(),
                tk.Id2Punc(3).c_str(), tk.Id2Punc(4).c_str(), tk.Id2Punc(5).c_str(), (int)tk.IsPunc(tk.Id2Punc(3)), (int)tk.IsPunc("x"));
    return 0;
  }
  if (cmd == "hotwords" && argc >= 4) {
    pfhip_host::HostVocab vocab;
    if (!vocab.Load(argv[2])) { std::fprintf(stderr, "cannot open tokens\n"); return 1; }
    pfhip_host::SegDictHost dict;
    const bool have_dict = std::string(argv[3]) != "-" && dict.Load(argv[3]);
    std::string line;
    while (std::getline(std::cin, line)) {
      std::vector<int> mat, lens;
      pfhip_host::HotwordIdMatrix(line, have_dict ? &dict : nullptr, [&](const std::string& u) { return vocab.GetIdByToken(u); }, mat, lens);
      std::printf("%zu |", lens.size());
      for (int l : lens) std::printf(" %d", l);
      std::printf(" |");
      for (int v : mat) std::printf(" %d", v);
      std::printf("\n");
    }
    return 0;
  }
  if (cmd == "vocabtext" && argc >= 3) {
    pfhip_host::HostVocab vocab;
    if (!vocab.Load(argv[2])) { std::fprintf(stderr, "cannot open tokens\n"); return 1; }
    const std::string language = argc > 3 ? argv[3] : "";
    std::string line;
    while (std::getline(std::cin, line)) {
      std::vector<int> ids;
      std::stringstream ss(line);
      int v;
      while (ss >> v) ids.push_back(v);
      for (unsigned char c : vocab.Vector2StringV2(ids, language)) std::printf("%02x", c);
      std::printf("\n");
    }
    return 0;
  }
  if (cmd == "jsonstrings" && argc >= 3) {
    std::ifstream f(argv[2]);
    std::stringstream ss; ss << f.rdbuf();
    std::vector<std::string> out;
    if (!pfhip_host::ReadJsonStringArray(ss.str(), 0, out)) return 1;
    for (const std::string& s : out) {
      for (unsigned char c : s) std::printf("%02x", c);
      std::printf("\n");
    }
    return 0;
  }
  return 2;
}
