// tsan_pool.cpp — the host mirror's threading under ThreadSanitizer (`make -C contextsv_amd/csrc tsan`; CPU only):
//   * csvhost::HostPool::parallel_for sections back to back, from two caller threads at once (the run's two chains use the two pools),
//     nested use running inline, an exception thrown by an item;
//   * csvhost::WorkerThreads tickets (the lane / merge / split tasks of SVCaller::runResident);
//   * the threaded BGZF writer and the threaded block inflate of BamReader on the same file.
// Exit code 0 and no ThreadSanitizer report = pass. Reference counterpart: include/ThreadPool.h + htslib's hts_set_threads.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../contextsv_amd/csrc/host/bam_io.h"
#include "../../contextsv_amd/csrc/host/par.h"

int main()
{
    using namespace csvhost;
    // ---- parallel sections
    std::vector<long> out(10000);
    auto section = [&](int threads, long mul) {
        HostPool::instance().parallel_for(out.size(), threads, [&](size_t i) { out[i] = (long)i * mul; });
        long s = 0;
        for (long v : out) s += v;
        return s;
    };
    for (int rep = 0; rep < 200; rep++)
        if (section(rep % 5, rep) != (long)(out.size() - 1) * (long)out.size() / 2 * rep) { fprintf(stderr, "parallel_for: wrong sum\n"); return 1; }
    {   // two chains side by side, each on its own pool
        std::vector<long> a(5000), b(5000);
        std::thread t([&] { HostPool::second_pool_flag() = true; for (int r = 0; r < 100; r++) HostPool::instance().parallel_for(b.size(), 0, [&](size_t i) { b[i] += (long)i; }); });
        for (int r = 0; r < 100; r++) HostPool::instance().parallel_for(a.size(), 0, [&](size_t i) { a[i] += (long)i; });
        t.join();
        if (a != b) { fprintf(stderr, "two pools: different results\n"); return 1; }
    }
    {   // nested: the inner section runs inline
        std::atomic<long> n{0};
        HostPool::instance().parallel_for(16, 0, [&](size_t) { HostPool::instance().parallel_for(8, 0, [&](size_t) { n++; }); });
        if (n != 128) { fprintf(stderr, "nested parallel_for: %ld\n", n.load()); return 1; }
    }
    {   // an item throws: the first exception reaches the caller, the pool stays usable
        bool caught = false;
        try { HostPool::instance().parallel_for(64, 0, [&](size_t i) { if (i == 13) throw std::runtime_error("item 13"); }); } catch (const std::runtime_error &) { caught = true; }
        if (!caught || section(0, 1) != (long)(out.size() - 1) * (long)out.size() / 2) { fprintf(stderr, "exception path\n"); return 1; }
    }
    // ---- tickets
    {
        WorkerThreads &pool = WorkerThreads::instance();
        std::vector<WorkerThreads::Ticket> t;
        std::atomic<int> done{0};
        for (int k = 0; k < 12; k++) t.push_back(pool.start([&done] { usleep(200); done++; }));
        for (auto &x : t) pool.wait(x);
        if (done != 12) { fprintf(stderr, "tickets: %d\n", done.load()); return 1; }
    }
    // ---- threaded BGZF writer + threaded reader
    {
        const char *d = getenv("TMPDIR");
        const std::string path = std::string(d && *d ? d : "/tmp") + "/csv_tsan_" + std::to_string((long)getpid()) + ".bam";
        BamHeader h;
        h.text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:5000000\n";
        h.names = {"chr1"}; h.lens = {5000000};
        BamWriter w;
        if (!w.open(path, h, 1, 4)) { fprintf(stderr, "writer: %s\n", w.error().c_str()); return 1; }
        std::vector<uint32_t> cig(300);
        for (size_t k = 0; k < cig.size(); k++) cig[k] = (uint32_t)((1 + k % 50) << 4) | (uint32_t)(k % 3 == 1 ? 1 : k % 3 == 2 ? 2 : 0);
        for (int r = 0; r < 20000; r++) w.add(0, 100 + r * 200, 60, 0, "r" + std::to_string(r), cig.data(), (uint32_t)cig.size(), nullptr, 0, nullptr);
        if (!w.close()) { fprintf(stderr, "writer close: %s\n", w.error().c_str()); return 1; }
        BamReader rd;
        BamReadOptions o; o.threads = 4; o.want_qnames = true; o.window_blocks = 16;
        BamShard s;
        if (!rd.open(path) || !rd.loadIndex() || !rd.readContig("chr1", o, s) || s.n_reads() != 20000) { fprintf(stderr, "reader: %s (%lu records)\n", rd.error().c_str(), (unsigned long)s.n_reads()); return 1; }
        remove(path.c_str()); remove((path + ".bai").c_str());
    }
    printf("tsan_pool: ok\n");
    return 0;
}
