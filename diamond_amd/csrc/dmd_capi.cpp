// Error reporting and ABI version of libdiamond_hip.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/diamond_hip.h"

static thread_local char g_err[512] = "";

void dmd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dmd_last_error(void) { return g_err; }
extern "C" int dmd_abi_version(void) { return 11; }

// environment switches are cached by their readers (DmdEnvInt, dmd_common.h) and re-read after this call
static int g_env_generation = 0;
int dmd_env_generation() { return g_env_generation; }
extern "C" void dmd_reload_env(void) { ++g_env_generation; }
