// libmqgan_hip.so: error channel + ABI version.
#include "common.h"

namespace mq {
thread_local char g_err[512] = "";
}

extern "C" int mq_abi_version(void) { return 1; }
extern "C" const char *mq_last_error(void) { return mq::g_err; }
