/* plugin_host.c — loads a TempestSDR source plugin (dlopen + the ten
 * tsdrplugin_* symbols).  Replaces TempestSDR/src/TSDRPluginLoader.c:33-87 with
 * the same observable behaviour: RTLD_NOW, TSDR_INCOMPATIBLE_PLUGIN when the
 * object cannot be opened, TSDR_ERR_PLUGIN when a symbol is missing,
 * tsdrplugin_cleanup before dlclose. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "tsdr_host.h"

int plugin_host_load(plugin_host_t *p, const char *path)
{
    memset(p, 0, sizeof(*p));
    p->dl = dlopen(path, RTLD_NOW);
    if (!p->dl) {
        fprintf(stderr, "tsdr: cannot load plugin %s: %s\n", path, dlerror());
        return TSDR_INCOMPATIBLE_PLUGIN;
    }
    struct { const char *name; void **slot; } want[] = {
        {"tsdrplugin_getName", (void **)&p->getName},
        {"tsdrplugin_init", (void **)&p->init},
        {"tsdrplugin_setsamplerate", (void **)&p->setsamplerate},
        {"tsdrplugin_getsamplerate", (void **)&p->getsamplerate},
        {"tsdrplugin_setbasefreq", (void **)&p->setbasefreq},
        {"tsdrplugin_stop", (void **)&p->stop},
        {"tsdrplugin_setgain", (void **)&p->setgain},
        {"tsdrplugin_getlasterrortext", (void **)&p->getlasterrortext},
        {"tsdrplugin_readasync", (void **)&p->readasync},
        {"tsdrplugin_cleanup", (void **)&p->cleanup},
    };
    for (size_t i = 0; i < sizeof(want) / sizeof(want[0]); i++) {
        *want[i].slot = dlsym(p->dl, want[i].name);
        if (!*want[i].slot) {
            dlclose(p->dl);
            memset(p, 0, sizeof(*p));
            return TSDR_ERR_PLUGIN;
        }
    }
    p->readasync_raw = (int (*)(tsdrplugin_readasync_raw_function, void *))dlsym(p->dl, "tsdrplugin_readasync_raw"); /* optional */
    p->memory_stable = (int (*)(void))dlsym(p->dl, "tsdrplugin_memory_stable"); /* optional */
    p->loaded = 1;
    return TSDR_OK;
}

void plugin_host_close(plugin_host_t *p)
{
    if (!p->loaded) return;
    if (p->cleanup) p->cleanup();
    dlclose(p->dl);
    memset(p, 0, sizeof(*p));
}
