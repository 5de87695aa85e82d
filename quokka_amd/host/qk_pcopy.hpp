// qk_pcopy.hpp — amrex::FabArray::ParallelCopy / ParallelAdd between two box layouts with independent owners, for the C++17 host mirror: the data motion of
// FillPatchTwoLevels / average_down / YAFluxRegister::Reflux / RemakeLevel when the levels have their own DistributionMapping (reference
// src/simulation.hpp:1789-1858, :1949-1964, :1308, :1672-1685).  A thin owner of a qk_pcopy_plan (csrc/qk_amr_pcopy.hip) and its peer buffers; same
// wire protocol as the ghost exchange (quokka_amr_simulation.hpp fillBoundaryConditions): pack -> one send / recv pair per peer -> same-rank items ->
// wait -> unpack.  The counterpart of quokka_amd/amr.py ParallelCopy.
// EVERY rank runs every plan, with or without items of its own: the test transport counts collective steps (qk_comm.hpp).
#ifndef QK_HOST_PCOPY_HPP_
#define QK_HOST_PCOPY_HPP_

#include <vector>

#include "quokka_amr_simulation.hpp"

namespace qkhost
{

struct PcopyPlan {
	qk_pcopy_plan *h = nullptr;
	std::vector<int> peer;
	std::vector<void *> send, recv;
	std::vector<int64_t> nsend, nrecv;
	char const *name = "ParallelCopy"; // (QK_COMM_TRACE)

	// src / dst: the boxes of ALL ranks with their owners; src_nghost: source cells beyond the valid boxes that count (ring_only: only those);
	// dst_nghost: the destination boxes grown by it are wanted, minus holes[b] if given
	PcopyPlan(qk_geometry const &geom, std::vector<qk_box> const &src, std::vector<int> const &srcOwner, int src_nghost, bool src_ring_only, std::vector<qk_box> const &dst,
		  std::vector<int> const &dstOwner, int dst_nghost, std::vector<qk_box> const *holes, int ncomp)
	{
		AMREX_ALWAYS_ASSERT(src.size() == srcOwner.size() && dst.size() == dstOwner.size() && (holes == nullptr || holes->size() == dst.size()));
		qk_box const none{};
		int const nothing = 0;
		check(qk_pcopy_plan_create(Runtime::get().ctx, &geom, static_cast<int>(src.size()), src.empty() ? &none : src.data(), src.empty() ? &nothing : srcOwner.data(),
					   src_nghost, src_ring_only ? 1 : 0, static_cast<int>(dst.size()), dst.empty() ? &none : dst.data(), dst.empty() ? &nothing : dstOwner.data(),
					   dst_nghost, (holes == nullptr || holes->empty()) ? nullptr : holes->data(), ncomp, Comm::get().rank, &h),
		      "qk_pcopy_plan_create");
		int const np = qk_pcopy_plan_num_peers(h);
		for (int k = 0; k < np; ++k) {
			int r = 0;
			int64_t ns = 0, nr = 0;
			check(qk_pcopy_plan_peer(h, k, &r, &ns, &nr), "qk_pcopy_plan_peer");
			peer.push_back(r);
			nsend.push_back(ns);
			nrecv.push_back(nr);
			send.push_back(amrex::DeviceArena::get().alloc(std::max<size_t>(static_cast<size_t>(ns) * sizeof(double), 8)));
			recv.push_back(amrex::DeviceArena::get().alloc(std::max<size_t>(static_cast<size_t>(nr) * sizeof(double), 8)));
		}
	}
	PcopyPlan(PcopyPlan const &) = delete;
	auto operator=(PcopyPlan const &) -> PcopyPlan & = delete;
	~PcopyPlan()
	{
		for (void *p : send) {
			amrex::DeviceArena::get().free(p);
		}
		for (void *p : recv) {
			amrex::DeviceArena::get().free(p);
		}
		qk_pcopy_plan_destroy(h);
	}

	// components [scomp_src, scomp_src + ncomp) of `src` (this rank's boxes of the source list, in list order) to [scomp_dst, ...) of `dst`
	void operator()(amrex::MultiFab const &src, amrex::MultiFab &dst, int scomp_src = 0, int scomp_dst = 0, bool add = false)
	{
		hipStream_t const cs = Runtime::get().computeStream();
		for (size_t k = 0; k < peer.size(); ++k) {
			if (nsend[k] > 0) {
				check(qk_ParallelCopy_pack(h, cs, static_cast<int>(k), tab(src), scomp_src, static_cast<double *>(send[k])), "qk_ParallelCopy_pack");
			}
		}
		Comm::get().label = name;
		Comm::get().exchangeBegin(peer, send, nsend, recv, nrecv, sizeof(double), cs);
		Comm::get().label = nullptr;
		check(qk_ParallelCopy_local(h, cs, tab(src), tab(dst), scomp_src, scomp_dst, add ? 1 : 0), "qk_ParallelCopy_local");
		Comm::get().exchangeEnd(cs);
		for (size_t k = 0; k < peer.size(); ++k) {
			if (nrecv[k] > 0) {
				check(qk_ParallelCopy_unpack(h, cs, static_cast<int>(k), tab(dst), scomp_dst, static_cast<const double *>(recv[k]), add ? 1 : 0), "qk_ParallelCopy_unpack");
			}
		}
	}
};

} // namespace qkhost

#endif // QK_HOST_PCOPY_HPP_
