// multi_ranks.hpp -- run one task per rank on its own host thread and surface the FIRST failure in the caller's thread.
// Used by MultiEngine (multi.cpp: one engine + thread per device) and, without any device, by the CPU test of its error path
// (sdmi_selftest_rank_errors, tests/test_distributed_cpu.py).
#pragma once
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdmi.h"
#include "error.hpp"

namespace sdmi {

// f(rank) for rank 0 .. R-1: ranks 1 .. R-1 on new threads, rank 0 on the caller's.  Every rank runs to completion (or to its own
// exception) before anything is reported -- no rank is abandoned mid-flight -- then the lowest failing rank's error is rethrown as
// "<label(rank)>: <message>" with its status; `after_all`, if given, runs first (MultiEngine: drain every device's stream so that
// nothing still writes into the caller's buffers when the error reaches it).
template <class F>
inline void run_on_ranks(int R, const std::function<std::string(int)>& label, F&& f, const std::function<void()>& after_all = nullptr) {
    std::vector<std::string> msg((size_t)R);
    std::vector<int> status((size_t)R, SDMI_OK);
    auto body = [&](int r) {
        try { f(r); }
        catch (const Error& e) { status[(size_t)r] = e.status; msg[(size_t)r] = e.what(); }
        catch (const std::exception& e) { status[(size_t)r] = SDMI_ERR_INVALID; msg[(size_t)r] = e.what(); }
        catch (...) { status[(size_t)r] = SDMI_ERR_INVALID; msg[(size_t)r] = "unknown error"; }
    };
    std::vector<std::thread> th;
    for (int r = 1; r < R; ++r) th.emplace_back(body, r);
    body(0);
    for (auto& t : th) t.join();
    bool failed = false;
    for (int r = 0; r < R; ++r) failed = failed || status[(size_t)r] != SDMI_OK;
    if (failed && after_all) after_all();
    for (int r = 0; r < R; ++r)
        if (status[(size_t)r] != SDMI_OK) throw Error(status[(size_t)r], label(r) + ": " + msg[(size_t)r]);
}

}  // namespace sdmi
