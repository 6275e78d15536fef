// pano_rccl.hpp - the one exchange of the camera-sharded path ("a single RCCL gather over xGMI of the warped +
// weighted tiles onto rank 0") behind the C-ABI.  librccl.so is opened on first use, so that libpano_hip.so loads on machines
// and in processes that never shard; the function pointer types are taken from <rccl/rccl.h>, so a signature drift is a
// compile error, not a crash.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
#include <string>

namespace pano {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;  // why it is not available
    std::string path;   // the name it was opened by
    bool ok = false;

    static Rccl& get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, [] { r.load(); });
        return r;
    }

  private:
    void load() {
        void* h = nullptr;
        // PANO_RCCL_LIB (read once, here): the library to open INSTEAD of the system's - a site build of RCCL, or the test
        // double of tests/src/fake_rccl.cpp that lets several ranks share one GPU.  No fallback from it: a path that does not
        // load is an error, not a reason to talk to another library behind the caller's back
        if (const char* forced = getenv("PANO_RCCL_LIB"); forced && *forced) {
            h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            path = forced;
        } else {
            for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) {
                    path = name;
                    break;
                }
            }
        }
        if (!h) {
            const char* e = dlerror();
            error = (path.empty() ? std::string("librccl.so") : path) + " not loadable: " + (e ? e : "?");
            return;
        }
        auto sym = [&](const char* n) { return dlsym(h, n); };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        ok = GetUniqueId && CommInitRank && CommDestroy && CommCount && GroupStart && GroupEnd && Send && Recv && GetErrorString;
        if (!ok) error = path + " lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclCommCount / ncclGroup* / ncclSend / ncclRecv";
    }
};

}  // namespace pano
