// abi.hip -- library identification entry points of libfrcnn_hip.so, and the tuning registry (frcnn_tune.h).
#include "frcnn_common.h"
#include "frcnn_tune.h"
#include <string.h>
#include <mutex>

extern char **environ;

namespace {
// A fixed table: keys and values are copied in, readers get a pointer into a slot.  frcnn_set_tuning must not run concurrently with a
// launch that reads the same key (documented in the header); concurrent readers are fine.
constexpr int kTuneSlots = 96, kTuneKey = 48, kTuneVal = 80;
struct TuneSlot { char key[kTuneKey]; char val[kTuneVal]; bool set; char val0[kTuneVal]; bool set0; };    // val0 / set0: the load-time snapshot
TuneSlot g_tune[kTuneSlots];
int g_tune_n = 0;
std::mutex g_tune_mu;
std::once_flag g_tune_once;

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
        i = g_tune_n++;
        strcpy(g_tune[i].key, k);
    }
    g_tune[i].set = val != nullptr;
    if (val) strcpy(g_tune[i].val, val);
    return FRCNN_OK;
}
// the one place the library looks at the process environment: a snapshot of the FRCNN_* variables at load time
void tune_snapshot_env() {
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "FRCNN_", 6) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (eq) tune_store(*e, (size_t)(eq - *e), eq + 1);
    }
}
void tune_init() {
    std::call_once(g_tune_once, [] {
        tune_snapshot_env();
        for (int i = 0; i < g_tune_n; ++i) { g_tune[i].set0 = g_tune[i].set; strcpy(g_tune[i].val0, g_tune[i].val); }
    });
}
}  // namespace

const char *frcnn_tune(const char *key) {
    tune_init();
    std::lock_guard<std::mutex> lock(g_tune_mu);
    const int i = tune_find(key);
    return i >= 0 && g_tune[i].set ? g_tune[i].val : nullptr;
}

extern "C" {

int frcnn_abi_version(void) { return 22; }

int frcnn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    return e == hipSuccess ? n : -(1000 + (int)e);
}

int frcnn_set_tuning(const char *key, const char *value) {
    if (!key || strncmp(key, "FRCNN_", 6) != 0) return FRCNN_ERR_INVALID;
    tune_init();
    std::lock_guard<std::mutex> lock(g_tune_mu);
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
    std::lock_guard<std::mutex> lock(g_tune_mu);
    for (int i = 0; i < g_tune_n; ++i) { g_tune[i].set = g_tune[i].set0; strcpy(g_tune[i].val, g_tune[i].val0); }     // back to the load-time snapshot
    return FRCNN_OK;
}

}  // extern "C"
