// Test driver for velox_amd/cpp/VeloxStreamPosix.h.
//
//   stream_frontend proto <dir>             no GPU: the wire protocol of the reference server
//                                           (VeloxServer.cpp:67-157) over loopback — LIST, GET ranges,
//                                           clipping at the end of the file, empty answers.
//   stream_frontend hostile                 no GPU: a server that answers with more bytes than were asked for
//                                           (or LISTs a smaller size than it serves) must not make the client
//                                           allocate the announced frame nor the session write past its buffer.
//   stream_frontend play <dir> <copies>     GPU: <copies> concurrent sessions per hosted track download
//                                           256 KiB ranges and decode 4096 samples per scheduler tick, all
//                                           refills of a tick in ONE GPU batch; one session per track seeks.
//                                           <dir>/<name>.vlx.expect holds the int64 samples the reference
//                                           decoder produces for that track (written by the Python test
//                                           from the oracle); every session must deliver exactly those.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>

#include "VeloxStreamPosix.h"

using namespace velox_stream;

static std::vector<uint8_t> slurp(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
#define CHECK(c, msg) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, msg); return 1; } } while (0)

static int proto(const std::string &dir)
{
    RangeServer srv;
    CHECK(srv.start(dir), "server start");
    RangeClient cl;
    CHECK(cl.connect_to("127.0.0.1", srv.port()), "connect");
    std::vector<TrackInfo> tracks;
    CHECK(cl.list(tracks) && tracks.size() == srv.track_count() && !tracks.empty(), "LIST");
    std::vector<uint8_t> got;
    for (const TrackInfo &t : tracks) {
        const std::vector<uint8_t> file = slurp(dir + "/" + t.name);
        CHECK(file.size() == t.size, "LIST size");
        CHECK(cl.get(t.id, 0, 100, got) && got.size() == std::min<size_t>(100, file.size()) && !memcmp(got.data(), file.data(), got.size()), "GET head");
        const uint32_t mid = t.size / 2;
        CHECK(cl.get(t.id, mid, 300000, got) && got.size() == std::min<size_t>(300000, t.size - mid) && !memcmp(got.data(), file.data() + mid, got.size()), "GET middle, clipped at the end");
        CHECK(cl.get(t.id, t.size - 1, 10, got) && got.size() == 1 && got[0] == file.back(), "GET last byte");
        CHECK(cl.get(t.id, t.size, 10, got) && got.empty(), "GET at the end is empty");
    }
    CHECK(cl.get((int)tracks.size(), 0, 10, got) && got.empty(), "GET of an unknown track is empty");
    CHECK(cl.get(-1, 0, 10, got) && got.empty(), "GET of a negative track is empty");
    // two clients at once, interleaved
    RangeClient cl2;
    CHECK(cl2.connect_to("127.0.0.1", srv.port()), "second connect");
    std::vector<uint8_t> a, b;
    CHECK(cl.get(0, 0, 64, a) && cl2.get(0, 0, 64, b) && a == b, "two clients");
    srv.stop();
    printf("PROTO OK tracks=%zu\n", tracks.size());
    return 0;
}

// A peer that ignores the requested length: every GET is answered with `serve_len` bytes (announced
// honestly in the frame header, or — `lie` — announced as 3 GiB with nothing behind it).
static int hostile()
{
    const int lfd = ::socket(AF_INET, SOCK_STREAM, 0);
    CHECK(lfd >= 0, "socket");
    sockaddr_in a{}; a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;
    socklen_t al = sizeof(a);
    CHECK(::bind(lfd, (sockaddr *)&a, sizeof(a)) == 0 && ::listen(lfd, 4) == 0 && getsockname(lfd, (sockaddr *)&a, &al) == 0, "bind");
    const int port = ntohs(a.sin_port);
    std::thread srv([lfd] {
        for (int conn = 0; conn < 2; conn++) {
            const int fd = ::accept(lfd, nullptr, nullptr);
            if (fd < 0) return;
            char buf[256];
            while (::recv(fd, buf, sizeof(buf) - 1, 0) > 0) {
                if (conn == 0) {                                          // 5000 bytes for whatever was asked
                    std::vector<uint8_t> junk(5000, 0xAB);
                    if (!detail::send_framed(fd, junk.data(), (uint32_t)junk.size())) break;
                } else {                                                  // a 3 GiB frame header and then silence
                    const uint32_t be = htonl(0xC0000000u);
                    if (!detail::send_all(fd, &be, 4)) break;
                }
            }
            ::close(fd);
        }
    });
    {
        RangeClient cl;
        CHECK(cl.connect_to("127.0.0.1", port), "connect");
        TrackInfo t{0, "x.vlx", 1000};                                    // "LIST" said 1000 bytes; the server sends 5000 per GET
        Session s(cl, t);
        CHECK(!s.download_some(), "an over-long answer must be refused");
        CHECK(s.downloaded() == 0, "nothing may be written from a refused answer");
        std::vector<uint8_t> out;
        CHECK(!cl.get(0, 0, 100, out) && out.empty(), "the connection is dropped after an over-answer");
    }
    {
        RangeClient cl;
        CHECK(cl.connect_to("127.0.0.1", port), "connect 2");
        std::vector<uint8_t> out;
        CHECK(!cl.get(0, 0, 4096, out) && out.empty(), "a 3 GiB frame for a 4 KiB request must be refused without allocating it");
    }
    ::shutdown(lfd, SHUT_RDWR); ::close(lfd);
    srv.join();
    printf("HOSTILE OK\n");
    return 0;
}

static int play(const std::string &dir, int copies)
{
    RangeServer srv;
    CHECK(srv.start(dir), "server start");
    RangeClient lister;
    CHECK(lister.connect_to("127.0.0.1", srv.port()), "connect");
    std::vector<TrackInfo> tracks;
    CHECK(lister.list(tracks) && !tracks.empty(), "LIST");
    std::map<int, std::vector<int64_t>> expect;
    for (const TrackInfo &t : tracks) {
        const std::vector<uint8_t> raw = slurp(dir + "/" + t.name + ".expect");
        std::vector<int64_t> v(raw.size() / 8);
        memcpy(v.data(), raw.data(), v.size() * 8);
        expect[t.id] = std::move(v);
    }
    std::vector<std::unique_ptr<RangeClient>> conns;
    std::vector<std::unique_ptr<Session>> owned;
    std::vector<Session *> sessions;
    std::vector<int> track_of;
    for (int c = 0; c < copies; c++)
        for (const TrackInfo &t : tracks) {
            conns.emplace_back(new RangeClient());
            CHECK(conns.back()->connect_to("127.0.0.1", srv.port()), "session connect");
            owned.emplace_back(new Session(*conns.back(), t));
            sessions.push_back(owned.back().get());
            track_of.push_back(t.id);
        }
    std::map<Session *, std::vector<int64_t>> got;
    std::map<Session *, size_t> index;
    for (size_t i = 0; i < sessions.size(); i++) index[sessions[i]] = i;
    Scheduler sched;
    size_t ticks = 0, batches = 0, idle = 0;
    bool sought = false;
    auto sink = [&](Session &s, const velox_sample_t *v, const uint8_t *, size_t n) { got[&s].insert(got[&s].end(), v, v + n); };
    for (;;) {
        bool all = true;
        for (Session *s : sessions) all = all && s->finished();
        if (all) break;
        const size_t d = sched.tick(sessions, sink);
        batches += d; ticks++;
        idle = d ? 0 : idle + 1;
        if (idle >= 100000) {
            for (Session *s : sessions)
                if (!s->finished()) fprintf(stderr, "stuck: track %d downloaded=%zu delivered=%zu total=%llu decoder=%d\n", track_of[index[s]],
                                            s->downloaded(), s->delivered(), (unsigned long long)s->info().total_samples, s->decoder() != nullptr);
            CHECK(false, "scheduler makes no progress");
        }
        // once, mid-way: the first copy of every track jumps back to a point inside its second chunk
        if (!sought && ticks == 6) {
            sought = true;
            for (size_t i = 0; i < tracks.size() && i < sessions.size(); i++) {
                Session *s = sessions[i];
                const size_t target = std::min<size_t>(8192 + 1001, s->info().total_samples);
                const bool can = s->decoder() && got[s].size() >= target;
                const bool ok = can && s->seek(target);
                if (getenv("VLX_TRACE")) fprintf(stderr, "seek session %zu: had %zu, target %zu, can %d ok %d err '%s'\n", i, got[s].size(), target, can, ok, VeloxCodec::LastError().c_str());
                if (ok) got[s].resize(target);                            // what was delivered beyond the target is replayed
            }
        }
    }
    for (Session *s : sessions) {
        const std::vector<int64_t> &want = expect[track_of[index[s]]];
        if (got[s].size() != want.size())
            fprintf(stderr, "session %zu track %d: got %zu samples, want %zu (downloaded %zu, delivered %zu, last error '%s')\n", index[s],
                    track_of[index[s]], got[s].size(), want.size(), s->downloaded(), s->delivered(), VeloxCodec::LastError().c_str());
    }
    for (Session *s : sessions) {
        const std::vector<int64_t> &want = expect[track_of[index[s]]];
        CHECK(s->info().total_samples == want.size(), "total_samples");
        CHECK(got[s].size() == want.size(), "sample count");
        CHECK(!memcmp(got[s].data(), want.data(), want.size() * 8), "samples differ from the reference decode");
    }
    srv.stop();
    printf("PLAY OK sessions=%zu ticks=%zu batches=%zu\n", sessions.size(), ticks, batches);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "proto")) return proto(argv[2]);
    if (argc >= 2 && !strcmp(argv[1], "hostile")) return hostile();
    if (argc >= 4 && !strcmp(argv[1], "play")) return play(argv[2], atoi(argv[3]));
    fprintf(stderr, "usage: stream_frontend proto <dir> | hostile | play <dir> <copies>\n");
    return 2;
}
