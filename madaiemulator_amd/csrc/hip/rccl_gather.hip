// rccl_gather.hip -- the ONE collective of the path (BASELINE.json north_star: "one GPU per shard with
// a single RCCL gather over xGMI at the end"): an all-gather of a few doubles per rank between the processes of a
// multi-process run of the host layer (csrc/host/ranks.c: PCA components or restart runs dealt to ranks, one GPU each).
// It replaces the mutex-guarded arg-max / the serial component loop of the reference's single process
// (libEmu/estimate_threaded.c:308-313, multivar_support.c:20-28) at the point where independent shards meet.
//
// librccl is opened at run time (dlopen, the copy beside this library's libamdhip64): a single-GPU user of libgpemu_hip.so
// needs no RCCL.  The ncclUniqueId travels from
// rank 0 to the others through a file in a directory all ranks can see (written under a temporary name and renamed, so a
// reader never sees half of it); one communicator per call -- the gather happens once, at the end of a search.
#include "gpemu_internal.hpp"

#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <time.h>
#include <unistd.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
	void *h = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static bool load_rccl(Rccl &r, std::string &err)
{
	// The RCCL that belongs to THIS library's HIP runtime: the librccl next to the libamdhip64 we are linked with, by full path.
	// A bare "librccl.so.1" would be answered with whatever copy the process already holds -- in a Python process that has
	// imported torch, torch's own librccl, which drives torch's own HIP runtime and sees none of our devices or streams.
	Dl_info self;
	if (dladdr((void *)&hipGetDeviceCount, &self) && self.dli_fname) {
		std::string dir(self.dli_fname);
		const size_t slash = dir.rfind('/');
		if (slash != std::string::npos) {
			dir.resize(slash + 1);
			r.h = dlopen((dir + "librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
			if (!r.h) r.h = dlopen((dir + "librccl.so").c_str(), RTLD_NOW | RTLD_LOCAL);
		}
	}
	const char *names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
	for (const char *n : names) {
		if (r.h) break;
		r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
	}
	if (!r.h) { err = std::string("cannot open librccl: ") + dlerror(); return false; }
	r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
	r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
	r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
	r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
	r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
	if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { err = "librccl lacks an entry point"; return false; }
	return true;
}

static bool write_id(const char *path, const ncclUniqueId &id)
{
	const std::string tmp = std::string(path) + ".tmp";
	FILE *f = fopen(tmp.c_str(), "wb");
	if (!f) return false;
	const bool ok = fwrite(&id, sizeof id, 1, f) == 1;
	fclose(f);
	return ok && rename(tmp.c_str(), path) == 0;
}

static bool read_id(const char *path, ncclUniqueId &id, double timeout_s)
{
	struct timespec t0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (;;) {
		FILE *f = fopen(path, "rb");
		if (f) {
			const bool ok = fread(&id, sizeof id, 1, f) == 1;
			fclose(f);
			if (ok) return true;
		}
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s) return false;
		usleep(2000);
	}
}

} // namespace

// recv[r * count + i] = rank r's send[i]; host buffers.  id_path: a file name all ranks agree on and can reach (rank 0
// creates it; use a fresh name per gather).  Returns GPEMU_OK, GPEMU_ERR_ARG, GPEMU_ERR_NO_DEVICE or GPEMU_ERR_HIP (the
// message goes to errbuf when given).
extern "C" int gpemu_rccl_allgather(int device, int rank, int world, const char *id_path, const double *send, int count,
                                    double *recv, char *errbuf, size_t errlen)
{
	auto fail = [&](int code, const std::string &msg) {
		if (errbuf && errlen) snprintf(errbuf, errlen, "%s", msg.c_str());
		return code;
	};
	if (world < 1 || rank < 0 || rank >= world || count < 1 || !send || !recv || !id_path) return fail(GPEMU_ERR_ARG, "bad argument");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GPEMU_ERR_NO_DEVICE, "no HIP device");
	if (device < 0 || device >= ndev) return fail(GPEMU_ERR_ARG, "bad device");
	Rccl R;
	std::string err;
	if (!load_rccl(R, err)) return fail(GPEMU_ERR_HIP, err);
	if (hipSetDevice(device) != hipSuccess) return fail(GPEMU_ERR_HIP, "hipSetDevice failed");
	(void)hipGetLastError();        // RCCL reads the thread's last HIP error after its launches: do not hand it a stale one
	ncclUniqueId id;
	memset(&id, 0, sizeof id);
	if (rank == 0) {
		const ncclResult_t e = R.GetUniqueId(&id);
		if (e != ncclSuccess) return fail(GPEMU_ERR_HIP, std::string("ncclGetUniqueId: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"));
		if (!write_id(id_path, id)) return fail(GPEMU_ERR_ARG, std::string("cannot write ") + id_path);
	} else if (!read_id(id_path, id, getenv("GPEMU_RCCL_WAIT_S") ? atof(getenv("GPEMU_RCCL_WAIT_S")) : 600.0)) {
		return fail(GPEMU_ERR_ARG, std::string("rank 0 never wrote ") + id_path);
	}
	ncclComm_t comm = nullptr;
	ncclResult_t e = R.CommInitRank(&comm, world, id, rank);
	if (e != ncclSuccess) return fail(GPEMU_ERR_HIP, std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"));
	hipStream_t st = nullptr;
	double *dsend = nullptr, *drecv = nullptr;
	hipError_t h = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	if (h == hipSuccess) h = hipMalloc(&dsend, (size_t)count * sizeof(double));
	if (h == hipSuccess) h = hipMalloc(&drecv, (size_t)count * world * sizeof(double));
	if (h == hipSuccess) h = hipMemcpyAsync(dsend, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st);
	int rc = GPEMU_OK;
	std::string msg;
	if (h != hipSuccess) { rc = GPEMU_ERR_HIP; msg = std::string("device buffers: ") + hipGetErrorString(h); }
	if (rc == GPEMU_OK) {
		e = R.AllGather(dsend, drecv, (size_t)count, ncclDouble, comm, st);
		if (e != ncclSuccess) { rc = GPEMU_ERR_HIP; msg = std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"); }
	}
	if (rc == GPEMU_OK) {
		h = hipMemcpyAsync(recv, drecv, (size_t)count * world * sizeof(double), hipMemcpyDeviceToHost, st);
		if (h == hipSuccess) h = hipStreamSynchronize(st);
		if (h != hipSuccess) { rc = GPEMU_ERR_HIP; msg = std::string("gather results: ") + hipGetErrorString(h); }
	}
	R.CommDestroy(comm);
	if (dsend) hipFree(dsend);
	if (drecv) hipFree(drecv);
	if (st) hipStreamDestroy(st);
	if (rank == 0) unlink(id_path);               // (every rank has joined the communicator: the id has been read)
	return rc == GPEMU_OK ? GPEMU_OK : fail(rc, msg);
}
