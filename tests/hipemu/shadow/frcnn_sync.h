// Emulator stand-in for csrc/frcnn_sync.h: workgroups run one after another on one OS thread, so every store
// is already visible; the ticket is a plain increment.  (Found first on the include path by build_emu.py.)
#pragma once
#include <hip/hip_runtime.h>
static inline void frcnn_drain_vmem() { hipemu::dma_wait(0); }
static inline void frcnn_release_agent() {}
static inline void frcnn_acquire_agent() {}
static inline int frcnn_ticket(int *counter) { int o = *counter; *counter = o + 1; return o; }
static inline void frcnn_counter_reset(int *counter) { *counter = 0; }
