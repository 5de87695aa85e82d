// qk_grid_layout.hpp — boxes -> ranks for the refined levels of the C++17 host mirror when every level has its own DistributionMapping (reference
// src/simulation.hpp:1421-1500, :1657-1702 hand AMRSimulation a BoxArray + DistributionMapping per level that AMReX made):
//   maxSize         amrex::BoxArray::maxSize(chunk)
//   chopGrids       amrex::AmrMesh::ChopGrids (refine_grid_layout = 1, AMReX's default, which the reference does not change)
//   distributeSfc   amrex::DistributionMapping::SFCProcessorMap, AMReX's default strategy
// AMReX is not vendored under /root/reference: these restate its documented behaviour (unpinned, like the rest of grid generation) and are the SAME
// functions as quokka_amd/amr_simulation.py max_size / chop_grids / distribute_sfc — tests/test_amr_grids.py compares the two on the same inputs.
// Plain C++ (no HIP): BoxT is anything with int lo[3], hi[3].
#ifndef QK_HOST_GRID_LAYOUT_HPP_
#define QK_HOST_GRID_LAYOUT_HPP_

#include <algorithm>
#include <array>
#include <cstdint>
#include <numeric>
#include <utility>
#include <vector>

namespace qkhost
{

// every box longer than chunk[d] is cut into ceil(len / chunk) nearly equal pieces (in units of the blocking factor where it divides the edge), x fastest
template <class BoxT> auto maxSize(std::vector<BoxT> const &boxes, std::array<int, 3> const &chunk, int blocking_factor) -> std::vector<BoxT>
{
	std::vector<BoxT> out;
	for (auto const &bx : boxes) {
		std::vector<std::pair<int, int>> cuts[3];
		for (int d = 0; d < 3; ++d) {
			int const n = bx.hi[d] - bx.lo[d] + 1;
			if (n <= chunk[d]) {
				cuts[d].emplace_back(bx.lo[d], bx.hi[d]);
				continue;
			}
			int const unit = (n % blocking_factor == 0) ? blocking_factor : 1;
			int const m = n / unit, nb = (n + chunk[d] - 1) / chunk[d];
			int const base = m / nb, rem = m % nb;
			int a = bx.lo[d];
			for (int i = 0; i < nb; ++i) {
				int const ln = (base + (i < rem ? 1 : 0)) * unit;
				cuts[d].emplace_back(a, a + ln - 1);
				a += ln;
			}
		}
		for (auto const &kz : cuts[2]) {
			for (auto const &ky : cuts[1]) {
				for (auto const &kx : cuts[0]) {
					BoxT b = bx;
					b.lo[0] = kx.first;
					b.hi[0] = kx.second;
					b.lo[1] = ky.first;
					b.hi[1] = ky.second;
					b.lo[2] = kz.first;
					b.hi[2] = kz.second;
					out.push_back(b);
				}
			}
		}
	}
	return out;
}

// while a level has fewer boxes than `target`, halve the chunk size — the longest direction first (ties: the highest dimension) — as long as the
// halved size is a multiple of the blocking factor, and re-apply maxSize
template <class BoxT>
auto chopGrids(std::vector<BoxT> boxes, int target, int max_grid_size, int blocking_factor, std::array<int, 3> const &domain_len, int ndim) -> std::vector<BoxT>
{
	std::array<int, 3> chunk{};
	for (int d = 0; d < 3; ++d) {
		chunk[d] = (d < ndim) ? std::min(max_grid_size, domain_len[d]) : 1;
	}
	while (static_cast<int>(boxes.size()) < target) {
		auto const prev = chunk;
		std::vector<int> order(ndim);
		std::iota(order.begin(), order.end(), 0);
		std::sort(order.begin(), order.end(), [&](int a, int b) { return (prev[a] != prev[b]) ? prev[a] > prev[b] : a > b; });
		for (int const d : order) {
			int const half = chunk[d] / 2;
			if (static_cast<int>(boxes.size()) < target && half > 0 && half % blocking_factor == 0) {
				chunk[d] = half;
				boxes = maxSize(boxes, chunk, blocking_factor);
			}
		}
		if (chunk == prev) {
			break;
		}
	}
	return boxes;
}

inline auto mortonIndex(int i, int j, int k) -> std::uint64_t
{
	std::uint64_t m = 0;
	for (int bit = 0; bit < 20; ++bit) {
		m |= (static_cast<std::uint64_t>((i >> bit) & 1) << (3 * bit)) | (static_cast<std::uint64_t>((j >> bit) & 1) << (3 * bit + 1)) |
		     (static_cast<std::uint64_t>((k >> bit) & 1) << (3 * bit + 2));
	}
	return m;
}

// the boxes in Morton order of their low corner (in units of `unit` cells) are cut into nranks contiguous runs of about equal volume (a run that its
// last box pushed over the mean gives that box back), and the runs are dealt, heaviest first, to the ranks in order of how little they already hold
// (rank_load: cells of the coarser levels; empty: nothing) — a level with fewer boxes than ranks lands on the least loaded ranks.  Deterministic and
// identical on every rank.
template <class BoxT> auto distributeSfc(std::vector<BoxT> const &boxes, int nranks, std::vector<long long> const &rank_load, int unit) -> std::vector<int>
{
	int const n = static_cast<int>(boxes.size());
	std::vector<int> owner(static_cast<size_t>(n), 0);
	if (nranks == 1) {
		return owner;
	}
	std::vector<long long> vol(static_cast<size_t>(n));
	std::vector<std::uint64_t> key(static_cast<size_t>(n));
	long long sum = 0;
	for (int b = 0; b < n; ++b) {
		vol[b] = 1;
		int c[3];
		for (int d = 0; d < 3; ++d) {
			vol[b] *= boxes[b].hi[d] - boxes[b].lo[d] + 1;
			int const x = boxes[b].lo[d];
			c[d] = (x >= 0) ? x / unit : -((-x + unit - 1) / unit); // (floor division)
		}
		key[b] = mortonIndex(c[0], c[1], c[2]);
		sum += vol[b];
	}
	std::vector<int> order(static_cast<size_t>(n));
	std::iota(order.begin(), order.end(), 0);
	std::sort(order.begin(), order.end(), [&](int a, int b) { return (key[a] != key[b]) ? key[a] < key[b] : a < b; });
	double const per = static_cast<double>(sum) / nranks;
	std::vector<std::vector<int>> runs;
	int K = 0;
	double total = 0.0;
	for (int i = 0; i < nranks; ++i) {
		std::vector<int> run;
		double v = 0.0;
		while (K < n && (i == nranks - 1 || v < per)) {
			v += static_cast<double>(vol[order[K]]);
			run.push_back(order[K]);
			++K;
		}
		total += v;
		if (total / (i + 1) > per && run.size() > 1 && i < nranks - 1) {
			--K;
			total -= static_cast<double>(vol[run.back()]);
			run.pop_back();
		}
		runs.push_back(std::move(run));
	}
	std::vector<long long> load(static_cast<size_t>(nranks), 0);
	for (size_t r = 0; r < rank_load.size() && r < load.size(); ++r) {
		load[r] = rank_load[r];
	}
	std::vector<int> ranks(static_cast<size_t>(nranks)), heavy(static_cast<size_t>(nranks));
	std::iota(ranks.begin(), ranks.end(), 0);
	std::iota(heavy.begin(), heavy.end(), 0);
	std::sort(ranks.begin(), ranks.end(), [&](int a, int b) { return (load[a] != load[b]) ? load[a] < load[b] : a < b; }); // LeastUsedCPUs
	std::vector<long long> runVol(static_cast<size_t>(nranks), 0);
	for (int i = 0; i < nranks; ++i) {
		for (int const b : runs[i]) {
			runVol[i] += vol[b];
		}
	}
	std::sort(heavy.begin(), heavy.end(), [&](int a, int b) { return (runVol[a] != runVol[b]) ? runVol[a] > runVol[b] : a < b; });
	for (int z = 0; z < nranks; ++z) {
		for (int const b : runs[heavy[z]]) {
			owner[b] = ranks[z];
		}
	}
	return owner;
}

} // namespace qkhost

#endif // QK_HOST_GRID_LAYOUT_HPP_
