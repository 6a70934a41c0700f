// rccl_gather.hip -- the ONE collective of the path (BASELINE.json north_star: "one GPU per shard with
// a single RCCL gather over xGMI at the end"): an all-gather of a few doubles per rank between the processes of a
// multi-process run of the host layer (csrc/host/ranks.c: PCA components or restart runs dealt to ranks, one GPU each).
// It replaces the mutex-guarded arg-max / the serial component loop of the reference's single process
// (libEmu/estimate_threaded.c:308-313, multivar_support.c:20-28) at the point where independent shards meet.
//
// Three steps, so that the ranks can meet when they START and not when the slowest of them has finished its training:
//   gpemu_rccl_unique_id    rank 0 makes the ncclUniqueId (128 bytes; the caller carries it to the other ranks)
//   gpemu_rccl_comm_create  every rank joins the communicator (ncclCommInitRank: the ranks' rendezvous)
//   gpemu_rccl_comm_allgather / gpemu_rccl_comm_destroy   the gather itself on the communicator's own stream; the end
// gpemu_rccl_allgather is the three in one call with the id travelling through a file (a gather between ranks that are
// known to arrive together).
//
// librccl is opened at run time (dlopen, the copy beside this library's libamdhip64, once per process): a single-GPU user
// of libgpemu_hip.so needs no RCCL, and building this file needs no RCCL headers either -- the five entry points and the few
// types they take are declared below as rccl.h declares them (rccl.h:40-43 ncclUniqueId, :467 ncclDouble = 8).
#include "gpemu_internal.hpp"

#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <time.h>
#include <unistd.h>

namespace {

// the part of rccl.h this file uses
typedef struct { char internal[GPEMU_RCCL_ID_BYTES]; } UniqueId;      // ncclUniqueId
typedef void *Comm;                                                   // ncclComm_t
typedef int Result;                                                   // ncclResult_t, ncclSuccess = 0
constexpr int kDouble = 8;                                            // ncclDataType_t ncclFloat64 / ncclDouble

struct Rccl {
	void *h = nullptr;
	Result (*GetUniqueId)(UniqueId *) = nullptr;
	Result (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
	Result (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
	Result (*CommDestroy)(Comm) = nullptr;
	const char *(*GetErrorString)(Result) = nullptr;
	std::string err;
};

static void load_rccl_once(Rccl &r)
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
	if (!r.h) { const char *e = dlerror(); r.err = std::string("cannot open librccl: ") + (e ? e : "?"); return; }
	r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
	r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
	r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
	r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
	r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
	if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) r.err = "librccl lacks an entry point";
}

// one handle per process (the library stays loaded: its communicators' threads may outlive any one call)
static Rccl *rccl(std::string &err)
{
	static Rccl R;
	static std::once_flag once;
	std::call_once(once, load_rccl_once, std::ref(R));
	if (!R.err.empty()) { err = R.err; return nullptr; }
	return &R;
}

static std::string rccl_error(const Rccl *R, const char *what, Result e)
{
	return std::string(what) + ": " + (R->GetErrorString ? R->GetErrorString(e) : "error");
}

struct CommState {
	Comm comm = nullptr;
	int device = 0, rank = 0, world = 1;
	hipStream_t stream = nullptr;
};

static int fail(char *errbuf, size_t errlen, int code, const std::string &msg)
{
	if (errbuf && errlen) snprintf(errbuf, errlen, "%s", msg.c_str());
	return code;
}

static bool write_id(const char *path, const UniqueId &id)
{
	const std::string tmp = std::string(path) + ".tmp";
	FILE *f = fopen(tmp.c_str(), "wb");
	if (!f) return false;
	const bool ok = fwrite(&id, sizeof id, 1, f) == 1;
	fclose(f);
	return ok && rename(tmp.c_str(), path) == 0;
}

static bool read_id(const char *path, UniqueId &id, double timeout_s)
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

extern "C" int gpemu_rccl_unique_id(void *id_out, char *errbuf, size_t errlen)
{
	if (!id_out) return fail(errbuf, errlen, GPEMU_ERR_ARG, "bad argument");
	std::string err;
	Rccl *R = rccl(err);
	if (!R) return fail(errbuf, errlen, GPEMU_ERR_HIP, err);
	UniqueId id;
	memset(&id, 0, sizeof id);
	const Result e = R->GetUniqueId(&id);
	if (e != 0) return fail(errbuf, errlen, GPEMU_ERR_HIP, rccl_error(R, "ncclGetUniqueId", e));
	memcpy(id_out, &id, sizeof id);
	return GPEMU_OK;
}

extern "C" int gpemu_rccl_comm_create(int device, int rank, int world, const void *id_in, void **comm_out, char *errbuf, size_t errlen)
{
	if (world < 1 || rank < 0 || rank >= world || !id_in || !comm_out) return fail(errbuf, errlen, GPEMU_ERR_ARG, "bad argument");
	*comm_out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(errbuf, errlen, GPEMU_ERR_NO_DEVICE, "no HIP device");
	if (device < 0 || device >= ndev) return fail(errbuf, errlen, GPEMU_ERR_ARG, "bad device");
	std::string err;
	Rccl *R = rccl(err);
	if (!R) return fail(errbuf, errlen, GPEMU_ERR_HIP, err);
	int cur = 0;
	(void)hipGetDevice(&cur);
	if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return fail(errbuf, errlen, GPEMU_ERR_HIP, "hipSetDevice failed"); }
	(void)hipGetLastError();        // RCCL reads the thread's last HIP error after its launches: do not hand it a stale one
	UniqueId id;
	memcpy(&id, id_in, sizeof id);
	CommState *S = new CommState();
	S->device = device; S->rank = rank; S->world = world;
	const Result e = R->CommInitRank(&S->comm, world, id, rank);
	if (e != 0) { delete S; (void)hipSetDevice(cur); return fail(errbuf, errlen, GPEMU_ERR_HIP, rccl_error(R, "ncclCommInitRank", e)); }
	if (hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking) != hipSuccess) {
		(void)hipGetLastError();
		R->CommDestroy(S->comm);
		delete S;
		(void)hipSetDevice(cur);
		return fail(errbuf, errlen, GPEMU_ERR_HIP, "hipStreamCreate failed");
	}
	(void)hipSetDevice(cur);
	*comm_out = S;
	return GPEMU_OK;
}

extern "C" int gpemu_rccl_comm_allgather(void *comm, const double *send, int count, double *recv, char *errbuf, size_t errlen)
{
	CommState *S = (CommState *)comm;
	if (!S || !send || !recv || count < 1) return fail(errbuf, errlen, GPEMU_ERR_ARG, "bad argument");
	std::string err;
	Rccl *R = rccl(err);
	if (!R) return fail(errbuf, errlen, GPEMU_ERR_HIP, err);
	int cur = 0;
	(void)hipGetDevice(&cur);
	if (hipSetDevice(S->device) != hipSuccess) { (void)hipGetLastError(); return fail(errbuf, errlen, GPEMU_ERR_HIP, "hipSetDevice failed"); }
	(void)hipGetLastError();
	double *dsend = nullptr, *drecv = nullptr;
	hipError_t h = hipMalloc(&dsend, (size_t)count * sizeof(double));
	if (h == hipSuccess) h = hipMalloc(&drecv, (size_t)count * S->world * sizeof(double));
	if (h == hipSuccess) h = hipMemcpyAsync(dsend, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, S->stream);
	int rc = GPEMU_OK;
	std::string msg;
	if (h != hipSuccess) { rc = GPEMU_ERR_HIP; msg = std::string("device buffers: ") + hipGetErrorString(h); }
	if (rc == GPEMU_OK) {
		const Result e = R->AllGather(dsend, drecv, (size_t)count, kDouble, S->comm, S->stream);
		if (e != 0) { rc = GPEMU_ERR_HIP; msg = rccl_error(R, "ncclAllGather", e); }
	}
	if (rc == GPEMU_OK) {
		h = hipMemcpyAsync(recv, drecv, (size_t)count * S->world * sizeof(double), hipMemcpyDeviceToHost, S->stream);
		if (h == hipSuccess) h = hipStreamSynchronize(S->stream);
		if (h != hipSuccess) { rc = GPEMU_ERR_HIP; msg = std::string("gather results: ") + hipGetErrorString(h); }
	}
	if (dsend) hipFree(dsend);
	if (drecv) hipFree(drecv);
	(void)hipGetLastError();
	(void)hipSetDevice(cur);
	return rc == GPEMU_OK ? GPEMU_OK : fail(errbuf, errlen, rc, msg);
}

extern "C" void gpemu_rccl_comm_destroy(void *comm)
{
	CommState *S = (CommState *)comm;
	if (!S) return;
	std::string err;
	Rccl *R = rccl(err);
	int cur = 0;
	(void)hipGetDevice(&cur);
	(void)hipSetDevice(S->device);
	if (R && S->comm) R->CommDestroy(S->comm);
	if (S->stream) hipStreamDestroy(S->stream);
	(void)hipSetDevice(cur);
	delete S;
}

// recv[r * count + i] = rank r's send[i]; host buffers.  id_path: a file name all ranks agree on and can reach (rank 0
// creates it; use a fresh name per gather).  Returns GPEMU_OK, GPEMU_ERR_ARG, GPEMU_ERR_NO_DEVICE or GPEMU_ERR_HIP (the
// message goes to errbuf when given).
extern "C" int gpemu_rccl_allgather(int device, int rank, int world, const char *id_path, const double *send, int count,
                                    double *recv, char *errbuf, size_t errlen)
{
	if (world < 1 || rank < 0 || rank >= world || count < 1 || !send || !recv || !id_path) return fail(errbuf, errlen, GPEMU_ERR_ARG, "bad argument");
	UniqueId id;
	memset(&id, 0, sizeof id);
	if (rank == 0) {
		const int rc = gpemu_rccl_unique_id(&id, errbuf, errlen);
		if (rc) return rc;
		if (!write_id(id_path, id)) return fail(errbuf, errlen, GPEMU_ERR_ARG, std::string("cannot write ") + id_path);
	} else if (!read_id(id_path, id, getenv("GPEMU_RCCL_WAIT_S") ? atof(getenv("GPEMU_RCCL_WAIT_S")) : 600.0)) {
		return fail(errbuf, errlen, GPEMU_ERR_ARG, std::string("rank 0 never wrote ") + id_path);
	}
	void *comm = nullptr;
	int rc = gpemu_rccl_comm_create(device, rank, world, &id, &comm, errbuf, errlen);
	if (rc == GPEMU_OK) {
		rc = gpemu_rccl_comm_allgather(comm, send, count, recv, errbuf, errlen);
		gpemu_rccl_comm_destroy(comm);
	}
	if (rank == 0) unlink(id_path);               // (every rank has joined the communicator: the id has been read)
	return rc;
}
