// abi.hip -- library identification entry points of libfrcnn_hip.so, and the tuning registry (frcnn_tune.h).
#include "frcnn_common.h"
#include "frcnn_tune.h"
#include <string.h>
#include <mutex>
#include <shared_mutex>
#include <vector>
#include <stdlib.h>

extern char **environ;

namespace {
// A fixed table of keys; VALUES ARE IMMUTABLE, interned strings: a set / reset swaps the slot's pointer and never rewrites or frees a
// string a reader may hold (frcnn_tune() used to hand out a pointer into a slot that frcnn_set_tuning / frcnn_reset_tuning
// strcpy'd over after the lock was gone).  The intern pool only grows by DISTINCT values (A/B knobs: a handful), so toggling a knob in
// a loop allocates nothing.  Readers share the lock (std::shared_mutex): launches on several threads do not serialise on it.
constexpr int kTuneSlots = 96, kTuneKey = 48, kTuneVal = 80;
struct TuneSlot { char key[kTuneKey]; const char *val; const char *val0; };      // val / val0 (the load-time snapshot): interned, nullptr = unset
TuneSlot g_tune[kTuneSlots];
int g_tune_n = 0;
std::shared_mutex g_tune_mu;
std::once_flag g_tune_once;
std::vector<char *> g_tune_pool;

const char *tune_intern(const char *v) {
    for (char *s : g_tune_pool)
        if (strcmp(s, v) == 0) return s;
    char *s = (char *)malloc(strlen(v) + 1);
    strcpy(s, v);
    g_tune_pool.push_back(s);
    return s;
}
int tune_find(const char *key) {
    for (int i = 0; i < g_tune_n; ++i)
        if (strcmp(g_tune[i].key, key) == 0) return i;
    return -1;
}
int tune_store(const char *key, size_t klen, const char *val) {
    if (klen == 0 || klen >= (size_t)kTuneKey || (val && strlen(val) >= (size_t)kTuneVal)) return FRCNN_ERR_INVALID;
    char k[kTuneKey];
    memcpy(k, key, klen); k[klen] = 0;
    int i = tune_find(k);
    if (i < 0) {
        if (!val) return FRCNN_OK;
        if (g_tune_n == kTuneSlots) return FRCNN_ERR_INVALID;
        i = g_tune_n;
        strcpy(g_tune[i].key, k);
        g_tune[i].val = g_tune[i].val0 = nullptr;
        g_tune_n = i + 1;
    }
    g_tune[i].val = val ? tune_intern(val) : nullptr;
    return FRCNN_OK;
}
// the one place the library looks at the process environment: a snapshot of the FRCNN_* variables at load time (entries the table
// cannot hold -- a 48+ character key, an 80+ character value, a 97th key -- are not knobs of this library and are skipped)
void tune_snapshot_env() {
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "FRCNN_", 6) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (eq) tune_store(*e, (size_t)(eq - *e), eq + 1);
    }
}
void tune_init() {
    std::call_once(g_tune_once, [] {
        std::unique_lock<std::shared_mutex> lock(g_tune_mu);
        tune_snapshot_env();
        for (int i = 0; i < g_tune_n; ++i) g_tune[i].val0 = g_tune[i].val;
    });
}
}  // namespace

const char *frcnn_tune(const char *key) {
    tune_init();
    std::shared_lock<std::shared_mutex> lock(g_tune_mu);
    const int i = tune_find(key);
    return i >= 0 ? g_tune[i].val : nullptr;         // immutable: valid for the life of the library, whatever is set afterwards
}

extern "C" {

int frcnn_abi_version(void) { return 24; }

int frcnn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    return e == hipSuccess ? n : -(1000 + (int)e);
}

int frcnn_set_tuning(const char *key, const char *value) {
    if (!key || strncmp(key, "FRCNN_", 6) != 0) return FRCNN_ERR_INVALID;
    tune_init();
    std::unique_lock<std::shared_mutex> lock(g_tune_mu);
    return tune_store(key, strlen(key), value);
}

int frcnn_get_tuning(const char *key, char *value_out, int capacity) {
    if (!key || capacity < 0 || (capacity > 0 && !value_out)) return FRCNN_ERR_INVALID;
    const char *v = frcnn_tune(key);
    if (!v) return 0;
    const int n = (int)strlen(v);
    if (capacity > 0) {
        const int c = n < capacity - 1 ? n : capacity - 1;
        memcpy(value_out, v, (size_t)c);
        value_out[c] = 0;
    }
    return n + 1;
}

int frcnn_reset_tuning(void) {
    tune_init();
    std::unique_lock<std::shared_mutex> lock(g_tune_mu);
    for (int i = 0; i < g_tune_n; ++i) g_tune[i].val = g_tune[i].val0;     // back to the load-time snapshot
    return FRCNN_OK;
}

}  // extern "C"
