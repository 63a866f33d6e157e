// frcnn_tune.h -- the library's tuning registry (round 5).
// Every A/B hook of the launchers reads this table, never the process environment: the table is filled ONCE, when the library is loaded,
// from the FRCNN_* variables present at that moment (abi.hip: the only getenv-like access in csrc/), and changed afterwards only through
// the ABI entry frcnn_set_tuning(key, value) (include/frcnn_hip.h).  A launch therefore costs no getenv, a concurrent setenv() in the host
// program cannot race a launch, and the knobs are visible in the header.  Defaults (= no entry) are the measured picks; no key selects a
// CPU path.
#pragma once
#include <stdlib.h>

// value of `key` or nullptr when unset.  The string is immutable and stays valid for the life of the library (a later set / reset swaps
// the slot to another interned string; it never rewrites this one).
const char *frcnn_tune(const char *key);
static inline int frcnn_tune_int(const char *key, int dflt) {
    const char *v = frcnn_tune(key);
    return v ? atoi(v) : dflt;
}
static inline bool frcnn_tune_is(const char *key, char first) {
    const char *v = frcnn_tune(key);
    return v && v[0] == first;
}
