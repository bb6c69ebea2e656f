// plugin.cpp -- Plugins/*.so loader (reference: Kernel/src/convnet/plugin.cu:40-70).  Every
// regular file in the plugin directory is dlopen()ed; its `layerConstructor` / `neuronConstructor`
// maps are merged into the registries ConvNet::initLayer, CostLayer::makeCostLayer and
// Neuron::makeNeuron fall back to.
#include <dirent.h>
#include <dlfcn.h>
#include <sys/stat.h>

#include <iostream>

#include "layer.h"

namespace rt {

std::map<std::string, layerConFunc>& pluginLayers() {
    static std::map<std::string, layerConFunc> m;
    return m;
}

std::map<std::string, neuronConFunc>& pluginNeurons() {
    static std::map<std::string, neuronConFunc> m;
    return m;
}

typedef std::map<std::string, layerConFunc> (*layerMapFunc)();
typedef std::map<std::string, neuronConFunc> (*neuronMapFunc)();

static int load_dir(const std::string& dirName) {
    DIR* dp = opendir(dirName.c_str());
    if (!dp) return -1;
    int loaded = 0;
    while (struct dirent* ent = readdir(dp)) {
        const std::string path = dirName + "/" + ent->d_name;
        struct stat st;
        if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
        if (path.size() < 3 || path.compare(path.size() - 3, 3, ".so") != 0) continue;
        void* h = dlopen(path.c_str(), RTLD_NOW);
        if (!h) {
            std::cout << "Cannot open library: " << dlerror() << std::endl;
            continue;
        }
        dlerror();
        layerMapFunc lf = (layerMapFunc)dlsym(h, "layerConstructor");
        neuronMapFunc nf = (neuronMapFunc)dlsym(h, "neuronConstructor");
        if (!lf && !nf) {
            std::cout << "Plugin " << path << " exports neither layerConstructor nor neuronConstructor" << std::endl;
            dlclose(h);
            continue;
        }
        if (lf)
            for (auto& kv : lf()) pluginLayers()[kv.first] = kv.second;
        if (nf)
            for (auto& kv : nf()) pluginNeurons()[kv.first] = kv.second;
        ++loaded;
    }
    closedir(dp);
    return loaded;
}

int loadPlugins(const char* moduleDir) {
    // next to the module first (so the package works from any working directory), then the
    // reference's CWD-relative "Plugins"
    int n = -1;
    if (moduleDir && *moduleDir) n = load_dir(std::string(moduleDir) + "/Plugins");
    if (n <= 0) {
        const int m = load_dir("Plugins");
        if (m > 0) n = m;
    }
    return n < 0 ? 0 : n;
}

}  // namespace rt
