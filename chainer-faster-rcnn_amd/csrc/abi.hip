// abi.hip -- library identification entry points of libfrcnn_hip.so.
#include "frcnn_common.h"

extern "C" {

int frcnn_abi_version(void) { return 21; }

int frcnn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    return e == hipSuccess ? n : -(1000 + (int)e);
}

}  // extern "C"
