// Leader/follower merging of concurrent callers into one batched device pass (offline forwards, streaming chunks, online-VAD
// calls).  The first caller to find no leader leads: it waits (bounded by a deadline and by `enough()`) for company, takes a
// batch off the queue, executes it outside the lock and hands each caller its result; then it promotes the next queued caller.
// Wake-ups are targeted — a push wakes only a waiting leader, completion wakes only the callers that were served — so a round
// of N callers costs O(N) wake-ups, not O(N^2) (128 handler threads on one condition variable spent more time waking each
// other than the device spent computing).
#pragma once
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <vector>

namespace pfhip_detail {

struct MergeReqBase {
  std::condition_variable cv;
  bool done = false, lead = false;
};

template <class Req>                     // Req : MergeReqBase
struct MergeQueue {
  std::mutex mu;
  std::condition_variable leader_cv;
  std::deque<Req*> q;
  bool leader_active = false, leader_waiting = false;

  // enough(q): stop waiting for company.  pick(q, take): move the batch from the queue into `take` (the front request — the
  // leader's own — must be taken).  exec(take): runs unlocked; must fill every request's result fields.
  template <class Enough, class Pick, class Exec>
  void submit(Req& me, int wait_us, Enough enough, Pick pick, Exec exec) {
    std::unique_lock<std::mutex> l(mu);
    q.push_back(&me);
    if (leader_waiting) leader_cv.notify_one();
    if (leader_active) {
      me.cv.wait(l, [&] { return me.done || me.lead; });
      if (me.done) return;
    } else {
      leader_active = true;
    }
    // leader; `me` is at the front of the queue
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us);
    leader_waiting = true;
    while (!enough(q) && leader_cv.wait_until(l, deadline) != std::cv_status::timeout) {}
    leader_waiting = false;
    std::vector<Req*> take;
    pick(q, take);
    l.unlock();
    exec(take);
    l.lock();
    for (Req* r : take) {
      r->done = true;
      if (r != &me) r->cv.notify_one();
    }
    if (!q.empty()) {
      q.front()->lead = true;
      q.front()->cv.notify_one();
    } else {
      leader_active = false;
    }
  }
};

}  // namespace pfhip_detail
