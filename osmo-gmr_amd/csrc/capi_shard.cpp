// capi_shard.cpp -- gmr1_hip_shard.h: the receive loop over the GPUs of one node, RCCL point-to-point exchanges.
#include "capi_common.h"

#include <dlfcn.h>

#include <chrono>
#include <vector>

#include <rccl/rccl.h>          // types and prototypes only: the functions are resolved at run time

#include <gmr1_hip_shard.h>

using namespace gmr1;

namespace {

struct Rccl {
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	bool ok = false;
};

const Rccl &rccl()
{
	static Rccl r = [] {
		Rccl x;
		// the process may already carry an RCCL (PyTorch ships its own): use that one, two copies must not meet
		void *h = RTLD_DEFAULT;
		if (!dlsym(h, "ncclSend")) {
			h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
			if (!h)
				h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
			if (!h)
				return x;
		}
#define GMR1_SYM(name) x.name = reinterpret_cast<decltype(x.name)>(dlsym(h, "nccl" #name))
		GMR1_SYM(GetUniqueId); GMR1_SYM(CommInitRank); GMR1_SYM(CommDestroy); GMR1_SYM(GroupStart); GMR1_SYM(GroupEnd);
		GMR1_SYM(Send); GMR1_SYM(Recv); GMR1_SYM(AllGather); GMR1_SYM(GetErrorString);
#undef GMR1_SYM
		x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Send && x.Recv &&
		       x.AllGather && x.GetErrorString;
		return x;
	}();
	return r;
}

#define RCCL_TRY(expr)                                                                       \
	do {                                                                                     \
		ncclResult_t r_ = (expr);                                                            \
		if (r_ != ncclSuccess)                                                               \
			return fail(-EIO, "%s: %s", #expr, rccl().GetErrorString(r_));                   \
	} while (0)

double ms_since(std::chrono::steady_clock::time_point t0)
{
	return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() / 1e6;
}

}  // namespace

struct gmr1_hip_shard {
	ncclComm_t comm = nullptr;
	bool own = false;
	int rank = 0, world = 1;
	// grow-only device staging: received carriers; records / counts on their way
	void *d_iq = nullptr;   size_t iq_bytes = 0;
	void *d_rec = nullptr;  size_t rec_bytes = 0;
	void *d_meta = nullptr; size_t meta_bytes = 0;
};

namespace {

int grow(void **p, size_t *have, size_t want)
{
	if (*have >= want)
		return 0;
	if (*p)
		HIP_TRY(hipFree(*p));
	*p = nullptr;
	*have = 0;
	HIP_TRY(hipMalloc(p, want));
	*have = want;
	return 0;
}

int make(struct gmr1_hip_shard **out, ncclComm_t comm, bool own, int rank, int world)
{
	gmr1_hip_shard *sh = new gmr1_hip_shard;
	sh->comm = comm; sh->own = own; sh->rank = rank; sh->world = world;
	*out = sh;
	return 0;
}

}  // namespace

extern "C" {

int gmr1_hip_shard_unique_id(void *id)
{
	if (!id)
		return fail(-EINVAL, "shard: NULL id");
	if (!rccl().ok)
		return fail(-ENOSYS, "shard: no RCCL in this process and librccl.so cannot be loaded");
	ncclUniqueId u;
	RCCL_TRY(rccl().GetUniqueId(&u));
	static_assert(sizeof(u) == GMR1_HIP_SHARD_ID_BYTES, "ncclUniqueId size");
	std::memcpy(id, &u, sizeof(u));
	return 0;
}

int gmr1_hip_shard_create(struct gmr1_hip_shard **out, const void *id, int rank, int world)
{
	if (!out || !id || world < 1 || rank < 0 || rank >= world)
		return fail(-EINVAL, "shard: bad rank %d / world %d", rank, world);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (!rccl().ok)
		return fail(-ENOSYS, "shard: no RCCL in this process and librccl.so cannot be loaded");
	ncclUniqueId u;
	std::memcpy(&u, id, sizeof(u));
	ncclComm_t comm = nullptr;
	RCCL_TRY(rccl().CommInitRank(&comm, world, u, rank));
	return make(out, comm, true, rank, world);
}

int gmr1_hip_shard_adopt(struct gmr1_hip_shard **out, void *nccl_comm, int rank, int world)
{
	if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world)
		return fail(-EINVAL, "shard: bad communicator / rank %d / world %d", rank, world);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (!rccl().ok)
		return fail(-ENOSYS, "shard: no RCCL in this process and librccl.so cannot be loaded");
	return make(out, static_cast<ncclComm_t>(nccl_comm), false, rank, world);
}

void gmr1_hip_shard_destroy(struct gmr1_hip_shard *sh)
{
	if (!sh)
		return;
	if (sh->d_iq) (void)hipFree(sh->d_iq);
	if (sh->d_rec) (void)hipFree(sh->d_rec);
	if (sh->d_meta) (void)hipFree(sh->d_meta);
	if (sh->own && sh->comm && rccl().ok)
		(void)rccl().CommDestroy(sh->comm);
	delete sh;
}

int gmr1_hip_rx_run_sharded(struct gmr1_hip_shard *sh, void *stream, int root, int n_arfcn, int sps, const float *iq,
                            const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                            struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                            int32_t *status, int32_t *n_chains, float *timing_ms)
{
	if (!sh || n_arfcn < 0 || !offset || !length || root < 0 || root >= sh->world)
		return fail(-EINVAL, "rx_run_sharded: bad argument");
	const int rank = sh->rank, world = sh->world;
	const bool is_root = rank == root;
	if (is_root && (!iq || !out || !n_records || max_records < 0))
		return fail(-EINVAL, "rx_run_sharded: root needs iq, out and n_records");
	hipStream_t st = (hipStream_t)stream;
	const Rccl &R = rccl();
	const int per = (n_arfcn + world - 1) / world;          // carriers a rank owns at most: a = rank, rank + world, ...
	auto t0 = std::chrono::steady_clock::now();

	// ---- scatter: carrier a -> rank a mod world, all transfers in one group --------------------------------------
	std::vector<int> mine;
	for (int a = rank; a < n_arfcn; a += world)
		mine.push_back(a);
	const int nm = (int)mine.size();
	std::vector<uint64_t> loff((size_t)nm), llen((size_t)nm);
	std::vector<uint16_t> lname((size_t)nm);
	const float *liq = iq;
	if (is_root) {
		for (int k = 0; k < nm; k++) { loff[k] = offset[mine[k]]; llen[k] = length[mine[k]]; }   // root's own stay where they are
	} else {
		uint64_t tot = 0;
		for (int k = 0; k < nm; k++) { loff[k] = tot; llen[k] = length[mine[k]]; tot += llen[k]; }
		int r = grow(&sh->d_iq, &sh->iq_bytes, (size_t)(tot ? tot : 1) * 8);
		if (r) return r;
		liq = static_cast<const float *>(sh->d_iq);
	}
	for (int k = 0; k < nm; k++)
		lname[k] = arfcn ? arfcn[mine[k]] : (uint16_t)mine[k];
	if (world > 1) {
		RCCL_TRY(R.GroupStart());
		if (is_root) {
			for (int a = 0; a < n_arfcn; a++)
				if (a % world != root && length[a])
					RCCL_TRY(R.Send(iq + 2 * offset[a], (size_t)length[a] * 2, ncclFloat, a % world, sh->comm, st));
		} else {
			for (int k = 0; k < nm; k++)
				if (llen[k])
					RCCL_TRY(R.Recv(static_cast<float *>(sh->d_iq) + 2 * loff[k], (size_t)llen[k] * 2, ncclFloat, root,
					                sh->comm, st));
		}
		RCCL_TRY(R.GroupEnd());
	}
	if (timing_ms) {
		HIP_TRY(hipStreamSynchronize(st));
		timing_ms[0] = (float)ms_since(t0);
		t0 = std::chrono::steady_clock::now();
	}

	// ---- the receive loop on this rank's carriers ----------------------------------------------------------------
	// a carrier yields at most ~22 frames per second of capture; the buffer is sized from the longest one
	uint64_t longest = 1;
	for (int k = 0; k < nm; k++)
		if (llen[k] > longest) longest = llen[k];
	const int cap_per = (int)(longest / ((uint64_t)sps * 39 * 24) + 64);      // one record per TDMA frame at the very most
	const int cap = nm > 0 ? nm * cap_per : 1;
	std::vector<gmr1_hip_rx_record> rec((size_t)cap);
	std::vector<int32_t> lstat((size_t)per, 0), lnch((size_t)per, 0);
	int nrec = 0;
	if (nm > 0) {
		std::vector<uint16_t> lidx((size_t)nm);
		for (int k = 0; k < nm; k++)
			lidx[k] = (uint16_t)k;
		int r = gmr1_hip_rx_run_dev(stream, nm, sps, liq, loff.data(), llen.data(), lidx.data(), rec.data(), cap, &nrec,
		                            lstat.data(), lnch.data());
		if (r) return r;
		if (nrec > cap)
			return fail(-EIO, "rx_run_sharded: %d records from %d carriers do not fit %d", nrec, nm, cap);
	}
	if (timing_ms) {
		timing_ms[1] = (float)ms_since(t0);
		t0 = std::chrono::steady_clock::now();
	}

	// ---- gather ---------------------------------------------------------------------------------------------------
	// per rank: [records of local carrier 0..per-1 | status | chains | total], all gathered everywhere (3 per + 1 words)
	const int mw = 3 * per + 1;
	std::vector<int32_t> meta((size_t)mw, 0), all((size_t)mw * world, 0);
	// (the loop labelled its records with the local carrier index: counted per carrier here, then given the caller's labels)
	for (int i = 0; i < nrec; i++) {
		const int k = rec[i].arfcn;
		meta[k]++;
		rec[i].arfcn = lname[k];
	}
	for (int k = 0; k < nm; k++) { meta[per + k] = lstat[k]; meta[2 * per + k] = lnch[k]; }
	meta[3 * per] = nrec;
	if (world > 1) {
		int r = grow(&sh->d_meta, &sh->meta_bytes, (size_t)mw * 4 * (world + 1));
		if (r) return r;
		int32_t *d_me = static_cast<int32_t *>(sh->d_meta), *d_all = d_me + mw;
		HIP_TRY(hipMemcpyAsync(d_me, meta.data(), (size_t)mw * 4, hipMemcpyHostToDevice, st));
		RCCL_TRY(R.AllGather(d_me, d_all, (size_t)mw, ncclInt32, sh->comm, st));
		HIP_TRY(hipMemcpyAsync(all.data(), d_all, (size_t)mw * 4 * world, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	} else {
		all = meta;
	}
	// records: every rank's block to root
	std::vector<size_t> base((size_t)world + 1, 0);
	for (int r = 0; r < world; r++)
		base[r + 1] = base[r] + (size_t)all[(size_t)r * mw + 3 * per];
	const size_t total = base[world];
	std::vector<gmr1_hip_rx_record> gathered;
	if (world > 1) {
		const size_t need = (is_root ? total : (size_t)nrec) * sizeof(gmr1_hip_rx_record);
		int r = grow(&sh->d_rec, &sh->rec_bytes, need ? need : 1);
		if (r) return r;
		unsigned char *d = static_cast<unsigned char *>(sh->d_rec);
		if (!is_root && nrec)
			HIP_TRY(hipMemcpyAsync(d, rec.data(), (size_t)nrec * sizeof(gmr1_hip_rx_record), hipMemcpyHostToDevice, st));
		RCCL_TRY(R.GroupStart());
		if (is_root) {
			for (int r2 = 0; r2 < world; r2++) {
				const size_t n = base[r2 + 1] - base[r2];
				if (r2 != root && n)
					RCCL_TRY(R.Recv(d + base[r2] * sizeof(gmr1_hip_rx_record), n * sizeof(gmr1_hip_rx_record), ncclUint8, r2,
					                sh->comm, st));
			}
		} else if (nrec) {
			RCCL_TRY(R.Send(d, (size_t)nrec * sizeof(gmr1_hip_rx_record), ncclUint8, root, sh->comm, st));
		}
		RCCL_TRY(R.GroupEnd());
		if (is_root) {
			gathered.resize(total ? total : 1);
			if (total)
				HIP_TRY(hipMemcpyAsync(gathered.data(), d, total * sizeof(gmr1_hip_rx_record), hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			// root's own block did not travel
			for (int i = 0; i < nrec; i++)
				gathered[base[root] + i] = rec[i];
		} else {
			HIP_TRY(hipStreamSynchronize(st));
		}
	} else {
		gathered.assign(rec.begin(), rec.begin() + nrec);
	}
	if (is_root) {
		// carrier order: carrier a is local carrier a / world of rank a mod world; inside a carrier the owner's order
		std::vector<size_t> cur(base.begin(), base.end() - 1);
		int n_out = 0;
		for (int a = 0; a < n_arfcn; a++) {
			const int r2 = a % world, k = a / world;
			const int32_t *m = &all[(size_t)r2 * mw];
			for (int i = 0; i < m[k]; i++, n_out++)
				if (n_out < max_records)
					out[n_out] = gathered[cur[r2] + i];
			cur[r2] += (size_t)m[k];
			if (status) status[a] = m[per + k];
			if (n_chains) n_chains[a] = m[2 * per + k];
		}
		*n_records = n_out;
	}
	if (timing_ms)
		timing_ms[2] = (float)ms_since(t0);
	return 0;
}

}  // extern "C"
