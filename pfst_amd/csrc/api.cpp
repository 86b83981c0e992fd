// Process-wide error string + ABI version of libpfst_hip.so.
#include <stdio.h>
#include <string.h>
#include "../../include/pfst_hip.h"

static char g_err[512] = "no error";

void pfst_set_error(const char* file, int line, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", file, line, msg);
}

extern "C" const char* pfst_last_error(void) { return g_err; }
extern "C" int pfst_abi_version(void) { return 1; }
