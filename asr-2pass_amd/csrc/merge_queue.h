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
  // fresh_no_wait: a caller that finds the queue idle (nobody leading, nothing executing) runs at once — a lone caller pays
  // nothing; company that arrives while it executes is gathered by the next leader, which has already waited that long.
  template <class Enough, class Pick, class Exec>
  void submit(Req& me, int wait_us, Enough enough, Pick pick, Exec exec, bool fresh_no_wait = false) {
    std::unique_lock<std::mutex> l(mu);
    q.push_back(&me);
    if (leader_waiting) leader_cv.notify_one();
    bool fresh = false;
    if (leader_active) {
      me.cv.wait(l, [&] { return me.done || me.lead; });
      if (me.done) return;
    } else {
      leader_active = true;
      fresh = true;
    }
    // leader; `me` is at the front of the queue
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(fresh && fresh_no_wait ? 0 : wait_us);
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

// The offline forward's queue: ONE queue in front of several execution slots (contexts that share one weight set, replicas on
// other devices).  Callers queue; the one at the front leads: it claims an idle slot, gathers company, takes a batch and runs it
// on that slot while the next caller in line already leads the next batch — so up to `slots` merged batches are in flight.
//   * nothing executing and nobody else queued: the leader runs at once (a lone caller pays no wait);
//   * every slot busy: the leader gathers for as long as it has to wait for a slot anyway (free);
//   * a slot idle while other batches execute: concurrency is evident, the leader waits up to wait_us for company.
// claim() / release(slot) run under `mu`; exec(slot, take) runs unlocked.
template <class Req, class Slot>
struct PoolQueue {
  std::mutex mu;
  std::condition_variable leader_cv;     // the one gathering leader waits here: arrivals, freed slots
  std::deque<Req*> q;
  bool gathering = false;                // q non-empty <=> a leader is gathering (it is q.front())
  int executing = 0;                     // batches taken and not yet finished

  void slot_freed() {                    // a slot became idle outside this queue's own exec (direct calls)
    std::lock_guard<std::mutex> l(mu);
    if (gathering) leader_cv.notify_one();
  }

  template <class Claim, class Release, class Enough, class Pick, class Exec>
  void submit(Req& me, int wait_us, Claim claim, Release release, Enough enough, Pick pick, Exec exec) {
    std::unique_lock<std::mutex> l(mu);
    q.push_back(&me);
    if (gathering) {
      leader_cv.notify_one();
      me.cv.wait(l, [&] { return me.done || me.lead; });
      if (me.done) return;
    } else {
      gathering = true;
    }
    // leader; `me` is q.front()
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us);
    Slot* slot = nullptr;
    for (;;) {
      if (!slot) slot = claim();
      if (!slot) { leader_cv.wait(l); continue; }                       // woken by arrivals and by freed slots
      if (enough(q)) break;
      if (executing == 0 && q.size() == 1) break;                        // alone on an idle device: go
      if (leader_cv.wait_until(l, deadline) == std::cv_status::timeout) break;
    }
    std::vector<Req*> take;
    pick(q, take);
    ++executing;
    if (!q.empty()) {
      q.front()->lead = true;
      q.front()->cv.notify_one();
    } else {
      gathering = false;
    }
    l.unlock();
    exec(slot, take);
    l.lock();
    --executing;
    release(slot);
    for (Req* r : take) {
      r->done = true;
      if (r != &me) r->cv.notify_one();
    }
    if (gathering) leader_cv.notify_one();
  }
};

}  // namespace pfhip_detail
