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
	void *d_rec = nullptr;  size_t rec_bytes = 0;    // this rank's records, as the receive loop left them (device)
	void *d_all = nullptr;  size_t all_bytes = 0;    // root: every rank's records
	void *d_meta = nullptr; size_t meta_bytes = 0;
	void *d_flag = nullptr;                       // 1 + world status words: "can every rank go on?" (allocated with the shard)
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
	if (hipMalloc(&sh->d_flag, (size_t)(world + 1) * 4) != hipSuccess) {
		delete sh;
		return fail(-ENOMEM, "shard: no device memory for the status block");
	}
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
	if (sh->d_all) (void)hipFree(sh->d_all);
	if (sh->d_meta) (void)hipFree(sh->d_meta);
	if (sh->d_flag) (void)hipFree(sh->d_flag);
	if (sh->own && sh->comm && rccl().ok)
		(void)rccl().CommDestroy(sh->comm);
	delete sh;
}

}  // extern "C"

namespace {

// A group of point-to-point calls must be closed whatever happens inside it: the first failure is kept, the rest of
// the group is still posted (a peer is waiting for every one of them), and ncclGroupEnd always runs.
struct Group {
	const Rccl &R;
	ncclResult_t first = ncclSuccess;
	const char *what = "";
	bool open = false;
	explicit Group(const Rccl &r) : R(r)
	{
		note(R.GroupStart(), "ncclGroupStart");
		open = first == ncclSuccess;
	}
	void note(ncclResult_t r, const char *w)
	{
		if (r != ncclSuccess && first == ncclSuccess) { first = r; what = w; }
	}
	int end()
	{
		if (open) {
			note(R.GroupEnd(), "ncclGroupEnd");
			open = false;
		}
		return first == ncclSuccess ? 0 : fail(-EIO, "%s: %s", what, R.GetErrorString(first));
	}
	~Group() { if (open) (void)R.GroupEnd(); }
};

// Collective "is everybody fine?": every rank contributes its status word; returns the first non-zero one in rank
// order (so every rank returns the same value), 0 when all are fine.  Uses the shard's small device block.
int agree(gmr1_hip_shard *sh, hipStream_t st, int32_t mine, int32_t *first_bad_rank)
{
	const Rccl &R = rccl();
	const int world = sh->world;
	*first_bad_rank = mine ? sh->rank : -1;
	if (world == 1)
		return mine;
	int32_t *d = static_cast<int32_t *>(sh->d_flag);
	std::vector<int32_t> all((size_t)world, 0);
	hipError_t e = hipMemcpyAsync(d, &mine, 4, hipMemcpyHostToDevice, st);
	ncclResult_t r = ncclSuccess;
	if (e == hipSuccess)
		r = R.AllGather(d, d + 1, 1, ncclInt32, sh->comm, st);
	if (e == hipSuccess && r == ncclSuccess)
		e = hipMemcpyAsync(all.data(), d + 1, (size_t)world * 4, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess && r == ncclSuccess)
		e = hipStreamSynchronize(st);
	if (e != hipSuccess || r != ncclSuccess)
		return fail(-EIO, "rx_run_sharded: status exchange failed (%s)", e != hipSuccess ? hipGetErrorString(e) : R.GetErrorString(r));
	for (int k = 0; k < world; k++)
		if (all[k]) { *first_bad_rank = k; return all[k]; }
	return 0;
}

int rx_run_sharded_impl(struct gmr1_hip_shard *sh, void *stream, int root, bool scatter, int n_arfcn, int sps, const float *iq,
                        const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                        struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                        int32_t *status, int32_t *n_chains, float *timing_ms)
{
	// ---- arguments every rank can check alike (same values everywhere by contract): a plain return is safe here ------
	if (!sh || n_arfcn < 0 || !offset || !length || root < 0 || root >= sh->world)
		return fail(-EINVAL, "rx_run_sharded: bad argument");
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "rx_run_sharded: sps=%d out of range (1..16)", sps);
	const int rank = sh->rank, world = sh->world;
	const bool is_root = rank == root;
	const int per = (n_arfcn + world - 1) / world;          // carriers a rank owns at most: a = rank, rank + world, ...
	if (per > 65535)
		return fail(-EINVAL, "rx_run_sharded: %d carriers per rank (records label their carrier with 16 bits)", per);
	// (make() allocates the status block the agreement points exchange on every rank, or fails on every rank alike)
	if (world > 1 && !sh->d_flag)
		return fail(-EIO, "rx_run_sharded: the shard has no status block");
	hipStream_t st = (hipStream_t)stream;
	const Rccl &R = rccl();
	int32_t bad_rank = -1;

	// ---- what only this rank can know: its own arguments and allocations.  From here on a local failure is CARRIED
	// to the next agreement point -- no rank leaves while the others are inside a collective ---------------------------
	int err = 0;
	if (is_root && (!out || !n_records || max_records < 0))
		err = fail(-EINVAL, "rx_run_sharded: root needs out and n_records");
	std::vector<int> mine;
	for (int a = rank; a < n_arfcn; a += world)
		mine.push_back(a);
	const int nm = (int)mine.size();
	if (!err && !iq && (scatter ? is_root : nm > 0))
		err = fail(-EINVAL, "rx_run_sharded: iq is NULL on a rank that holds samples");
	std::vector<uint64_t> loff((size_t)nm), llen((size_t)nm);
	std::vector<uint16_t> lname((size_t)nm);
	const float *liq = iq;
	if (!scatter || is_root) {
		for (int k = 0; k < nm; k++) { loff[k] = offset[mine[k]]; llen[k] = length[mine[k]]; }   // already where they are used
	} else {
		uint64_t tot = 0;
		for (int k = 0; k < nm; k++) { loff[k] = tot; llen[k] = length[mine[k]]; tot += llen[k]; }
		if (!err)
			err = grow(&sh->d_iq, &sh->iq_bytes, (size_t)(tot ? tot : 1) * 8);
		liq = static_cast<const float *>(sh->d_iq);
	}
	for (int k = 0; k < nm; k++)
		lname[k] = arfcn ? arfcn[mine[k]] : (uint16_t)mine[k];
	{
		const int rc = agree(sh, st, err, &bad_rank);
		if (rc)
			return err ? err : fail(rc, "rx_run_sharded: rank %d cannot take part (%d)", bad_rank, rc);
	}
	// (the clock of the scatter starts behind the first agreement: its all-gather and stream synchronise are no transfer)
	auto t0 = std::chrono::steady_clock::now();

	// ---- scatter: carrier a -> rank a mod world, all transfers in one group --------------------------------------
	if (scatter && world > 1) {
		Group g(R);
		if (is_root) {
			for (int a = 0; a < n_arfcn; a++)
				if (a % world != root && length[a])
					g.note(R.Send(iq + 2 * offset[a], (size_t)length[a] * 2, ncclFloat, a % world, sh->comm, st), "ncclSend (scatter)");
		} else {
			for (int k = 0; k < nm; k++)
				if (llen[k])
					g.note(R.Recv(static_cast<float *>(sh->d_iq) + 2 * loff[k], (size_t)llen[k] * 2, ncclFloat, root, sh->comm, st),
					       "ncclRecv (scatter)");
		}
		err = g.end();
	}
	if (timing_ms) {
		if (!err && hipStreamSynchronize(st) != hipSuccess)
			err = fail(-EIO, "rx_run_sharded: stream synchronise after the scatter failed");
		timing_ms[0] = (float)ms_since(t0);
		t0 = std::chrono::steady_clock::now();
	}

	// ---- the receive loop on this rank's carriers ----------------------------------------------------------------
	// A carrier yields about one record per TDMA frame and chain; FCCH acquisition keeps up to 16 chains (beams) per
	// carrier.  Start with room for two chains per carrier and run again with the real count if that was too little
	// (gmr1_hip_rx_run_dev reports the full count even when it could not store it).
	uint64_t longest = 1;
	for (int k = 0; k < nm; k++)
		if (llen[k] > longest) longest = llen[k];
	const size_t cap_per = (size_t)(longest / ((uint64_t)sps * 39 * 24) + 64);
	// The records stay on the device from the loop to the send: the loop closes them up in sh->d_rec (device memory is a
	// legal record buffer of gmr1_hip_rx_run_dev), labelled with the caller's carrier names, and reports how many each
	// carrier contributed; only counts travel through the host.
	std::vector<int32_t> lstat((size_t)per + 1, 0), lnch((size_t)per + 1, 0), lcnt((size_t)per + 1, 0);
	int nrec = 0;
	if (!err && nm > 0) {
		size_t cap = (size_t)nm * cap_per * 2;
		for (int attempt = 0; attempt < 2 && !err; attempt++) {
			if (cap > (size_t)INT32_MAX) cap = INT32_MAX;
			err = grow(&sh->d_rec, &sh->rec_bytes, cap * sizeof(gmr1_hip_rx_record));
			if (err)
				break;
			err = rx_run_dev_counted(stream, nm, sps, liq, loff.data(), llen.data(), lname.data(),
			                         static_cast<gmr1_hip_rx_record *>(sh->d_rec), (int)cap, &nrec, lstat.data(), lnch.data(),
			                         lcnt.data());
			if (err || (size_t)nrec <= cap)
				break;
			cap = (size_t)nrec;
			if (attempt == 1)
				err = fail(-EIO, "rx_run_sharded: the record count changed between two runs over the same samples");
		}
	}
	if (err)
		nrec = 0;
	if (timing_ms) {
		timing_ms[1] = (float)ms_since(t0);
		t0 = std::chrono::steady_clock::now();
	}

	// ---- gather ---------------------------------------------------------------------------------------------------
	// per rank: [records of local carrier 0..per-1 | status | chains | total | this rank's error], gathered everywhere
	const int mw = 3 * per + 2;
	std::vector<int32_t> meta((size_t)mw, 0), all((size_t)mw * world, 0);
	for (int k = 0; k < nm; k++) { meta[k] = err ? 0 : lcnt[k]; meta[per + k] = lstat[k]; meta[2 * per + k] = lnch[k]; }
	meta[3 * per] = nrec;
	if (world > 1) {
		// (an allocation failure here is carried in the status word like any other: the block below is then skipped by
		// this rank only if it cannot even hold the counts, which the d_flag agreement makes known to everybody first)
		int aerr = grow(&sh->d_meta, &sh->meta_bytes, (size_t)mw * 4 * (world + 1));
		if (!err) err = aerr;
		const int rc = agree(sh, st, aerr, &bad_rank);
		if (rc)
			return err ? err : fail(rc, "rx_run_sharded: rank %d could not stage its counts (%d)", bad_rank, rc);
		meta[3 * per + 1] = err;
		int32_t *d_me = static_cast<int32_t *>(sh->d_meta), *d_all = d_me + mw;
		hipError_t e = hipMemcpyAsync(d_me, meta.data(), (size_t)mw * 4, hipMemcpyHostToDevice, st);
		ncclResult_t r = R.AllGather(d_me, d_all, (size_t)mw, ncclInt32, sh->comm, st);    // posted whatever happened before
		if (e == hipSuccess && r == ncclSuccess)
			e = hipMemcpyAsync(all.data(), d_all, (size_t)mw * 4 * world, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess && r == ncclSuccess)
			e = hipStreamSynchronize(st);
		if (e != hipSuccess || r != ncclSuccess)
			return fail(-EIO, "rx_run_sharded: count exchange failed (%s)", e != hipSuccess ? hipGetErrorString(e) : R.GetErrorString(r));
	} else {
		meta[3 * per + 1] = err;
		all = meta;
	}
	// a rank whose receive loop failed has told everybody: all ranks stop here, with that rank's error
	for (int r2 = 0; r2 < world; r2++) {
		const int32_t e2 = all[(size_t)r2 * mw + 3 * per + 1];
		if (e2)
			return err ? err : fail(e2, "rx_run_sharded: the receive loop failed on rank %d (%d)", r2, e2);
	}
	// records: every rank's block to root, device to device (root's own block by a copy on its stream)
	std::vector<size_t> base((size_t)world + 1, 0);
	for (int r = 0; r < world; r++)
		base[r + 1] = base[r] + (size_t)all[(size_t)r * mw + 3 * per];
	const size_t total = base[world];
	std::vector<gmr1_hip_rx_record> gathered;
	{
		int aerr = 0;
		if (is_root)
			aerr = grow(&sh->d_all, &sh->all_bytes, (total ? total : 1) * sizeof(gmr1_hip_rx_record));
		if (world > 1) {
			const int rc = agree(sh, st, aerr, &bad_rank);
			if (rc)
				return aerr ? aerr : fail(rc, "rx_run_sharded: rank %d could not stage the records (%d)", bad_rank, rc);
		} else if (aerr) {
			return aerr;
		}
		unsigned char *d_mine = static_cast<unsigned char *>(sh->d_rec);
		unsigned char *d = static_cast<unsigned char *>(sh->d_all);
		hipError_t e = hipSuccess;
		if (is_root && nrec)
			e = hipMemcpyAsync(d + base[root] * sizeof(gmr1_hip_rx_record), d_mine, (size_t)nrec * sizeof(gmr1_hip_rx_record),
			                   hipMemcpyDeviceToDevice, st);
		if (world > 1) {
			Group g(R);
			if (is_root) {
				for (int r2 = 0; r2 < world; r2++) {
					const size_t n = base[r2 + 1] - base[r2];
					if (r2 != root && n)
						g.note(R.Recv(d + base[r2] * sizeof(gmr1_hip_rx_record), n * sizeof(gmr1_hip_rx_record), ncclUint8, r2,
						              sh->comm, st), "ncclRecv (records)");
				}
			} else if (nrec) {
				g.note(R.Send(d_mine, (size_t)nrec * sizeof(gmr1_hip_rx_record), ncclUint8, root, sh->comm, st), "ncclSend (records)");
			}
			err = g.end();
		}
		if (!err && e != hipSuccess)
			err = fail(-EIO, "rx_run_sharded: placing the root's own records failed (%s)", hipGetErrorString(e));
		if (is_root && !err) {
			gathered.resize(total ? total : 1);
			if (total && hipMemcpyAsync(gathered.data(), d, total * sizeof(gmr1_hip_rx_record), hipMemcpyDeviceToHost, st) != hipSuccess)
				err = fail(-EIO, "rx_run_sharded: reading the gathered records failed");
		}
		if (hipStreamSynchronize(st) != hipSuccess && !err)
			err = fail(-EIO, "rx_run_sharded: stream synchronise after the gather failed");
		if (err)
			return err;
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

}  // namespace

extern "C" {

int gmr1_hip_rx_run_sharded(struct gmr1_hip_shard *sh, void *stream, int root, int n_arfcn, int sps, const float *iq,
                            const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                            struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                            int32_t *status, int32_t *n_chains, float *timing_ms)
{
	return rx_run_sharded_impl(sh, stream, root, true, n_arfcn, sps, iq, offset, length, arfcn, out, max_records, n_records,
	                           status, n_chains, timing_ms);
}

int gmr1_hip_rx_run_sharded_resident(struct gmr1_hip_shard *sh, void *stream, int root, int n_arfcn, int sps, const float *iq,
                                     const uint64_t *offset, const uint64_t *length, const uint16_t *arfcn,
                                     struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                                     int32_t *status, int32_t *n_chains, float *timing_ms)
{
	return rx_run_sharded_impl(sh, stream, root, false, n_arfcn, sps, iq, offset, length, arfcn, out, max_records, n_records,
	                           status, n_chains, timing_ms);
}

}  // extern "C"
