// qk_comm.hpp — the multi-rank layer of the C++ host mirror: one process per GPU (reference src/main.cpp:22-46: amrex::Initialize over MPI, one rank
// per device), started by any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT (e.g.
// `python -m torch.distributed.run --no-python --nproc-per-node 8 bin/test_hydro3d_blast deck.in`).
//
// What travels between ranks, and nothing else:
//   * the ghost strips of FillBoundary: packed / unpacked by the library's kernels (qk_FillBoundary_pack / _unpack, the same plan every rank
//     builds from the global BoxArray and the box -> rank map), moved peer to peer;
//   * a handful of scalars per step (max signal speed, FOFC / retry counters, error flags, diagnostic sums): all-reduces.
// Transport "rccl" (production): ncclSend / ncclRecv of all peers in one group on a library-owned second HIP stream, ordered against the compute
// stream by events — the strips go GPU to GPU over xGMI, the host never touches them; all-reduces with ncclAllReduce on the same stream.
// Transport "shm" (QK_COMM_BACKEND=shm; tests on ONE GPU, where RCCL refuses two ranks on a device): the same protocol with the buffers staged
// through the host and exchanged as files under /dev/shm.  Everything above the transport — box distribution, plans, pack / unpack, ordering,
// reductions — is shared, which is what the one-GPU test exercises.
#ifndef QK_COMM_HPP_
#define QK_COMM_HPP_

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace qkhost
{

class Comm
{
      public:
	int rank = 0, size = 1, local_rank = 0;
	enum class Backend { none, rccl, shm } backend = Backend::none;

	static auto get() -> Comm &
	{
		static Comm c;
		return c;
	}

	// the RCCL stream (non-blocking), nullptr before init() on several ranks: what else must be waited for before fab memory is reused
	[[nodiscard]] auto commStream() const -> hipStream_t { return stream_; }

	// reads the launcher's environment, selects the device of this rank, opens the communicator (idempotent)
	void init()
	{
		if (initialised_) {
			return;
		}
		initialised_ = true;
		auto envInt = [](char const *name, int dflt) {
			char const *v = std::getenv(name);
			return (v != nullptr && *v != 0) ? std::atoi(v) : dflt;
		};
		rank = envInt("RANK", 0);
		size = envInt("WORLD_SIZE", 1);
		local_rank = envInt("LOCAL_RANK", rank);
		// rendezvous names are unique per launch: the launcher's port and its process id (the parent of every rank)
		char const *port = std::getenv("MASTER_PORT");
		tag_ = std::string("qkcomm_") + ((port != nullptr) ? port : "0") + "_" + std::to_string(static_cast<long>(getppid()));
		// QK_GHOST_LOOPBACK=1 on one rank (self-test of the production transport on ONE GPU): the ghost plan routes the pairs of this rank's own
		// boxes through a peer that is this rank itself (csrc/qk_boundary.hip) and the 1-rank communicator below moves them with grouped
		// ncclSend / ncclRecv to self on the communication stream
		loopback_ = size <= 1 && envInt("QK_GHOST_LOOPBACK", 0) != 0;
		if (size <= 1 && !loopback_) {
			size = 1;
			rank = 0;
			return;
		}
		if (loopback_) {
			size = 1;
			rank = 0;
			local_rank = 0;
		}
		char const *be = std::getenv("QK_COMM_BACKEND");
		backend = (be != nullptr && std::string(be) == "shm") ? Backend::shm : Backend::rccl;
		int ndev = 0;
		hipCheck(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
		if (backend == Backend::rccl) {
			if (local_rank >= ndev) {
				die("more ranks on this node than GPUs (RCCL needs one device per rank; QK_COMM_BACKEND=shm shares one device for tests)");
			}
			hipCheck(hipSetDevice(local_rank), "hipSetDevice");
			hipCheck(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
			hipCheck(hipEventCreateWithFlags(&evReady_, hipEventDisableTiming), "hipEventCreate");
			hipCheck(hipEventCreateWithFlags(&evDone_, hipEventDisableTiming), "hipEventCreate");
			ncclUniqueId id;
			std::string const path = "/tmp/" + tag_ + "_id";
			if (rank == 0) {
				ncclCheck(ncclGetUniqueId(&id), "ncclGetUniqueId");
				writeFile(path, &id, sizeof(id));
			} else {
				readFile(path, &id, sizeof(id));
			}
			ncclCheck(ncclCommInitRank(&nccl_, size, id, rank), "ncclCommInitRank");
			hipCheck(hipMalloc(reinterpret_cast<void **>(&d_scratch_), 64 * sizeof(double)), "hipMalloc");
			barrier();
			if (rank == 0) {
				std::remove(path.c_str());
			}
			if (loopback_) {
				std::printf("qkhost::Comm: QK_GHOST_LOOPBACK — ghost strips of this rank's own boxes travel through ncclSend / ncclRecv to self\n");
			}
		} else {
			hipCheck(hipSetDevice(local_rank % ndev), "hipSetDevice");
		}
	}

	~Comm()
	{
		if (nccl_ != nullptr) {
			ncclCommDestroy(nccl_);
		}
	}

	// ---- the ghost strips.  exchangeBegin moves, for every peer k, send[k] (device, nsend[k] elements of `elemBytes`) to peer[k] and receives
	// recv[k] from it: ordered after the producer kernels (pack) already launched on `compute`, running on the communication stream while
	// `compute` goes on (same-rank copies).  exchangeEnd orders `compute` after the receives: consumer kernels (unpack) launched afterwards see
	// the data.
	void exchangeEnd(hipStream_t compute)
	{
		if (backend == Backend::rccl && inFlight_) {
			hipCheck(hipStreamWaitEvent(compute, evDone_, 0), "hipStreamWaitEvent");
		}
		inFlight_ = false;
	}
	void exchangeBegin(std::vector<int> const &peer, std::vector<void *> const &send, std::vector<int64_t> const &nsend, std::vector<void *> const &recv,
		      std::vector<int64_t> const &nrecv, size_t elemBytes, hipStream_t compute)
	{
		if (size == 1 && !loopback_) {
			return;
		}
		if (backend == Backend::shm) {
			++seq_; // (every rank counts every collective step, with or without peers of its own)
		}
		if (trace_) { // QK_COMM_TRACE=1: every exchange of every rank on stderr (who diverged from whom)
			std::string line = "qkcomm[" + std::to_string(rank) + "] exchange " + std::to_string(seq_) + " (" + (label != nullptr ? label : "") + "):";
			for (size_t k = 0; k < peer.size(); ++k) {
				line += " " + std::to_string(peer[k]) + ":s" + std::to_string(nsend[k]) + "/r" + std::to_string(nrecv[k]);
			}
			std::fprintf(stderr, "%s\n", line.c_str());
		}
		if (peer.empty()) {
			return;
		}
		if (backend == Backend::rccl) {
			hipCheck(hipEventRecord(evReady_, compute), "hipEventRecord");
			hipCheck(hipStreamWaitEvent(stream_, evReady_, 0), "hipStreamWaitEvent");
			ncclCheck(ncclGroupStart(), "ncclGroupStart");
			for (size_t k = 0; k < peer.size(); ++k) { // (a ParallelCopy plan often moves data one way only: the empty direction is skipped on both sides —
				if (nsend[k] > 0) {		  // this rank's nrecv from a peer is that peer's nsend to this rank)
					ncclCheck(ncclSend(send[k], static_cast<size_t>(nsend[k]) * elemBytes, ncclChar, peer[k], nccl_, stream_), "ncclSend");
				}
				if (nrecv[k] > 0) {
					ncclCheck(ncclRecv(recv[k], static_cast<size_t>(nrecv[k]) * elemBytes, ncclChar, peer[k], nccl_, stream_), "ncclRecv");
				}
			}
			ncclCheck(ncclGroupEnd(), "ncclGroupEnd");
			hipCheck(hipEventRecord(evDone_, stream_), "hipEventRecord");
			inFlight_ = true;
			return;
		}
		// shm: stage through the host
		hipCheck(hipStreamSynchronize(compute), "hipStreamSynchronize");
		std::vector<char> h;
		for (size_t k = 0; k < peer.size(); ++k) {
			h.resize(static_cast<size_t>(nsend[k]) * elemBytes);
			hipCheck(hipMemcpy(h.data(), send[k], h.size(), hipMemcpyDeviceToHost), "hipMemcpy");
			writeFile(msgPath("x", rank, peer[k], seq_), h.data(), h.size());
		}
		for (size_t k = 0; k < peer.size(); ++k) {
			h.resize(static_cast<size_t>(nrecv[k]) * elemBytes);
			std::string const p = msgPath("x", peer[k], rank, seq_);
			readFile(p, h.data(), h.size());
			std::remove(p.c_str());
			hipCheck(hipMemcpy(recv[k], h.data(), h.size(), hipMemcpyHostToDevice), "hipMemcpy");
		}
	}

	// ---- scalars
	enum class Op { sum, max, min };
	void allReduce(double *v, int n, Op op)
	{
		if (size == 1) {
			return;
		}
		if (n > 64) {
			die("allReduce: at most 64 values");
		}
		if (backend == Backend::rccl) {
			hipCheck(hipMemcpyAsync(d_scratch_, v, sizeof(double) * n, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync");
			ncclRedOp_t const o = (op == Op::sum) ? ncclSum : (op == Op::max ? ncclMax : ncclMin);
			ncclCheck(ncclAllReduce(d_scratch_, d_scratch_, static_cast<size_t>(n), ncclDouble, o, nccl_, stream_), "ncclAllReduce");
			hipCheck(hipMemcpyAsync(v, d_scratch_, sizeof(double) * n, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync");
			hipCheck(hipStreamSynchronize(stream_), "hipStreamSynchronize");
			return;
		}
		++seq_;
		if (trace_) {
			std::fprintf(stderr, "qkcomm[%d] allReduce %ld (%d doubles)\n", rank, static_cast<long>(seq_), n);
		}
		writeFile(msgPath("r", rank, -1, seq_), v, sizeof(double) * n);
		// summation in rank order on every rank: all ranks obtain the same bits
		std::vector<std::vector<double>> all(static_cast<size_t>(size));
		for (int r = 0; r < size; ++r) {
			all[r].resize(static_cast<size_t>(n));
			if (r == rank) {
				std::memcpy(all[r].data(), v, sizeof(double) * n);
			} else {
				readFile(msgPath("r", r, -1, seq_), all[r].data(), sizeof(double) * n);
			}
		}
		for (int i = 0; i < n; ++i) {
			double a = all[0][i];
			for (int r = 1; r < size; ++r) {
				double const b = all[r][i];
				a = (op == Op::sum) ? a + b : (op == Op::max ? (a > b ? a : b) : (a < b ? a : b));
			}
			v[i] = a;
		}
		barrier(); // every rank has read every file
		std::remove(msgPath("r", rank, -1, seq_).c_str());
	}
	// any length, 64 values at a time (the per-fab minima / maxima of a MultiFab header written by rank 0)
	void allReduceMany(double *v, size_t n, Op op)
	{
		for (size_t i = 0; i < n; i += 64) {
			allReduce(v + i, static_cast<int>(std::min<size_t>(64, n - i)), op);
		}
	}
	// element-wise maximum of an int array of any length (the tile flags of a regrid: every rank clusters the same global flags)
	void allReduceMaxInts(int *v, size_t n)
	{
		if (size == 1 || n == 0) {
			return;
		}
		if (backend == Backend::rccl) {
			int *d = nullptr;
			hipCheck(hipMalloc(reinterpret_cast<void **>(&d), sizeof(int) * n), "hipMalloc");
			hipCheck(hipMemcpyAsync(d, v, sizeof(int) * n, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync");
			ncclCheck(ncclAllReduce(d, d, n, ncclInt, ncclMax, nccl_, stream_), "ncclAllReduce");
			hipCheck(hipMemcpyAsync(v, d, sizeof(int) * n, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync");
			hipCheck(hipStreamSynchronize(stream_), "hipStreamSynchronize");
			hipCheck(hipFree(d), "hipFree");
			return;
		}
		++seq_;
		if (trace_) {
			std::fprintf(stderr, "qkcomm[%d] allReduceMaxInts %ld (%zu ints)\n", rank, static_cast<long>(seq_), n);
		}
		writeFile(msgPath("r", rank, -1, seq_), v, sizeof(int) * n);
		std::vector<int> other(n);
		for (int r = 0; r < size; ++r) {
			if (r != rank) {
				readFile(msgPath("r", r, -1, seq_), other.data(), sizeof(int) * n);
				for (size_t i = 0; i < n; ++i) {
					v[i] = std::max(v[i], other[i]);
				}
			}
		}
		barrier(); // every rank has read every file
		std::remove(msgPath("r", rank, -1, seq_).c_str());
	}
	auto allReduceMax(double v) -> double
	{
		allReduce(&v, 1, Op::max);
		return v;
	}
	auto allReduceMin(double v) -> double
	{
		allReduce(&v, 1, Op::min);
		return v;
	}
	auto allReduceSum(double v) -> double
	{
		allReduce(&v, 1, Op::sum);
		return v;
	}
	// integers up to 2^53 exactly
	auto allReduceSum(int64_t v) -> int64_t { return static_cast<int64_t>(allReduceSum(static_cast<double>(v))); }
	auto allReduceMax(int v) -> int { return static_cast<int>(allReduceMax(static_cast<double>(v))); }

	void barrier()
	{
		if (size == 1) {
			return;
		}
		if (backend == Backend::rccl) {
			double z = 0;
			allReduce(&z, 1, Op::sum);
			return;
		}
		++seq_;
		char const one = 1;
		writeFile(msgPath("b", rank, -1, seq_), &one, 1);
		for (int r = 0; r < size; ++r) {
			char c = 0;
			readFile(msgPath("b", r, -1, seq_), &c, 1);
		}
		// files of barrier number s are removed when barrier s + 2 is entered: by then every rank has passed s
		if (seq_ > 2 && lastBarrier_[0] != 0) {
			std::remove(msgPath("b", rank, -1, lastBarrier_[0]).c_str());
		}
		lastBarrier_[0] = lastBarrier_[1];
		lastBarrier_[1] = seq_;
	}

	char const *label = nullptr; // what the next exchange is (QK_COMM_TRACE output only)

      private:
	bool trace_ = std::getenv("QK_COMM_TRACE") != nullptr;
	bool initialised_ = false;
	bool loopback_ = false;
	std::string tag_;
	ncclComm_t nccl_ = nullptr;
	hipStream_t stream_ = nullptr; // (non-blocking: amrex::DeviceArena asks for it — commStream() — before it lets a released block be reused)
	hipEvent_t evReady_ = nullptr, evDone_ = nullptr;
	double *d_scratch_ = nullptr;
	uint64_t seq_ = 0;
	uint64_t lastBarrier_[2] = {0, 0};
	bool inFlight_ = false;

	[[noreturn]] static void die(std::string const &msg)
	{
		std::fprintf(stderr, "qkhost::Comm: %s\n", msg.c_str());
		std::exit(3);
	}
	static void hipCheck(hipError_t e, char const *what)
	{
		if (e != hipSuccess) {
			die(std::string(what) + ": " + hipGetErrorString(e));
		}
	}
	static void ncclCheck(ncclResult_t r, char const *what)
	{
		if (r != ncclSuccess) {
			die(std::string(what) + ": " + ncclGetErrorString(r));
		}
	}
	[[nodiscard]] auto msgPath(char const *kind, int src, int dst, uint64_t seq) const -> std::string
	{
		return "/dev/shm/" + tag_ + "_" + kind + "_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(seq);
	}
	// whole-file hand-over: written under a temporary name, renamed when complete; the reader polls for the final name
	static void writeFile(std::string const &path, void const *data, size_t bytes)
	{
		std::string const tmp = path + ".tmp" + std::to_string(static_cast<long>(getpid()));
		{
			std::ofstream f(tmp, std::ios::binary);
			f.write(static_cast<char const *>(data), static_cast<std::streamsize>(bytes));
			if (!f) {
				die("cannot write " + tmp);
			}
		}
		if (std::rename(tmp.c_str(), path.c_str()) != 0) {
			die("cannot rename " + tmp);
		}
	}
	static auto waitSeconds() -> int // QK_COMM_TIMEOUT: how long a rank waits for a peer's message before it gives up (default 300 s)
	{
		static int const s = (std::getenv("QK_COMM_TIMEOUT") != nullptr) ? std::max(1, std::atoi(std::getenv("QK_COMM_TIMEOUT"))) : 300;
		return s;
	}
	static void readFile(std::string const &path, void *data, size_t bytes)
	{
		auto const t0 = std::chrono::steady_clock::now();
		struct stat st {
		};
		while (stat(path.c_str(), &st) != 0) {
			if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(waitSeconds())) {
				die("timed out waiting for " + path);
			}
			std::this_thread::sleep_for(std::chrono::microseconds(50));
		}
		std::ifstream f(path, std::ios::binary);
		f.read(static_cast<char *>(data), static_cast<std::streamsize>(bytes));
		if (f.gcount() != static_cast<std::streamsize>(bytes)) {
			die("short read of " + path);
		}
	}
};

} // namespace qkhost

#endif // QK_COMM_HPP_
