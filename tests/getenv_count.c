/* test helper (LD_PRELOAD): counts getenv("PDDC_...") calls between two marker calls.
 * getenv("__PDDC_WATCH_ON__") starts counting, "__PDDC_WATCH_OFF__" stops, "__PDDC_WATCH_COUNT__" returns the count as text. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

static int watching, count;
static char text[32];

char *getenv(const char *name)
{
    static char *(*real)(const char *);
    if (!real)
        real = (char *(*)(const char *))dlsym(RTLD_NEXT, "getenv");
    if (name && !strcmp(name, "__PDDC_WATCH_ON__")) {
        watching = 1;
        count = 0;
        return NULL;
    }
    if (name && !strcmp(name, "__PDDC_WATCH_OFF__")) {
        watching = 0;
        return NULL;
    }
    if (name && !strcmp(name, "__PDDC_WATCH_COUNT__")) {
        snprintf(text, sizeof(text), "%d", count);
        return text;
    }
    if (watching && name && !strncmp(name, "PDDC_", 5))
        ++count;
    return real ? real(name) : NULL;
}
