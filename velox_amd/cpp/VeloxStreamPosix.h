// VeloxStreamPosix.h — streaming front-end for BASELINE config 5 on POSIX sockets.
//
// Three pieces, all host-side plumbing around the GPU decoder of VeloxCodecAMD.h:
//
//   velox_stream::RangeServer   the reference server's wire protocol (VeloxServer.cpp:67-84 framing,
//                               :103-157 commands): a request is one ASCII line, "LIST" or
//                               "GET <track> <offset> <length>"; every answer is a big-endian u32
//                               byte count followed by that many bytes.  LIST answers
//                               "<id>|<file name>|<file size>\n" per hosted .vlx file; GET answers the
//                               byte range, clipped to the file, empty for a bad id or an offset past
//                               the end.  A reference client can talk to this server and this
//                               client to a reference server.
//   velox_stream::RangeClient   the matching client (VeloxStreamClient.cpp:472-513 issues the same
//                               256 KiB GETs).
//   velox_stream::Session       one track being played: a buffer of the whole file size that fills
//                               front to back, the container prologue parsed out of the first bytes
//                               (:355-377), and a StreamingDecoder over the payload part that is told
//                               how far the download has come (Extend).  Seeks go through the chunk
//                               walk (StreamingDecoder::Seek), not a restart plus fast-forward (:396-422).
//   velox_stream::Scheduler     many sessions, one GPU: per tick every session that has the bytes for
//                               its next chunk joins ONE refill batch (StreamingDecoder::RefillMany),
//                               then each delivers its next 4096 samples (:450) to its sink.
//
// Nothing here does codec arithmetic.
#ifndef VELOX_STREAM_POSIX_H
#define VELOX_STREAM_POSIX_H

#include <arpa/inet.h>
#include <dirent.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "VeloxCodecAMD.h"

namespace velox_stream {

struct TrackInfo { int id; std::string name; uint32_t size; };

namespace detail {
inline bool send_all(int fd, const void *p, size_t n)
{
    const char *c = static_cast<const char *>(p);
    while (n) {
        const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
        if (k <= 0) return false;
        c += k; n -= (size_t)k;
    }
    return true;
}
inline bool recv_all(int fd, void *p, size_t n)
{
    char *c = static_cast<char *>(p);
    while (n) {
        const ssize_t k = ::recv(fd, c, n, 0);
        if (k <= 0) return false;
        c += k; n -= (size_t)k;
    }
    return true;
}
// One answer: u32 length in network byte order, then the bytes (VeloxServer.cpp:67-84).
inline bool send_framed(int fd, const void *p, uint32_t n)
{
    const uint32_t be = htonl(n);
    return send_all(fd, &be, 4) && (n == 0 || send_all(fd, p, n));
}
} // namespace detail

class RangeServer {
    struct Hosted { std::string name, path; uint32_t size; };
    std::vector<Hosted> tracks_;
    int listen_fd_ = -1; int port_ = 0;
    std::atomic<bool> stop_{false};
    std::thread accept_thread_;
    // one entry per connection: the serving thread marks it done (and forgets the fd) before it closes the
    // socket, so stop() never shuts down a descriptor number the process has reused, and accept_loop joins
    // finished threads instead of keeping one dead std::thread per connection for ever
    struct Client { std::thread th; int fd; bool done; };
    std::mutex mu_; std::vector<std::unique_ptr<Client>> clients_;

    void serve(Client *self)
    {
        const int fd = self->fd;
        char buf[256];
        for (;;) {
            const ssize_t n = ::recv(fd, buf, sizeof(buf) - 1, 0);   // one command per recv, as the reference reads it
            if (n <= 0) break;
            buf[n] = 0;
            int id = -1; unsigned off = 0, len = 0;
            if (!strncmp(buf, "LIST", 4)) {
                std::string body;
                for (size_t i = 0; i < tracks_.size(); i++)
                    body += std::to_string(i) + "|" + tracks_[i].name + "|" + std::to_string(tracks_[i].size) + "\n";
                if (!detail::send_framed(fd, body.data(), (uint32_t)body.size())) break;
            } else if (sscanf(buf, "GET %d %u %u", &id, &off, &len) == 3) {
                std::vector<uint8_t> data;
                if (id >= 0 && (size_t)id < tracks_.size() && off < tracks_[id].size) {
                    if ((uint64_t)off + len > tracks_[id].size) len = tracks_[id].size - off;
                    if (FILE *f = fopen(tracks_[id].path.c_str(), "rb")) {
                        data.resize(len);
                        if (fseek(f, (long)off, SEEK_SET) != 0) data.clear();
                        else data.resize(fread(data.data(), 1, len, f));
                        fclose(f);
                    }
                }
                if (!detail::send_framed(fd, data.data(), (uint32_t)data.size())) break;
            } else if (!detail::send_framed(fd, nullptr, 0)) break;   // unknown command: empty answer
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            self->fd = -1; self->done = true;
        }
        ::close(fd);
    }
    void reap_locked()                                                    // join and drop the connections that have ended
    {
        for (size_t i = 0; i < clients_.size();) {
            if (clients_[i]->done) {
                if (clients_[i]->th.joinable()) clients_[i]->th.join();   // past its last use of mu_: joins at once
                clients_.erase(clients_.begin() + (long)i);
            } else i++;
        }
    }
    void accept_loop()
    {
        while (!stop_) {
            const int fd = ::accept(listen_fd_, nullptr, nullptr);
            if (fd < 0) { if (stop_) break; continue; }
            int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            std::lock_guard<std::mutex> g(mu_);
            reap_locked();
            clients_.emplace_back(new Client{std::thread(), fd, false});
            Client *cl = clients_.back().get();
            cl->th = std::thread([this, cl] { serve(cl); });
        }
    }

public:
    RangeServer() = default;
    RangeServer(const RangeServer &) = delete;
    ~RangeServer() { stop(); }

    // Host every *.vlx file of `dir` (sorted by name, so ids are stable) and listen on 127.0.0.1:port
    // (0 = any free port, see port()).  Returns false if nothing could be bound.
    bool start(const std::string &dir, int port = 0)
    {
        if (DIR *d = opendir(dir.c_str())) {
            while (dirent *e = readdir(d)) {
                const std::string n = e->d_name;
                if (n.size() < 4 || n.compare(n.size() - 4, 4, ".vlx")) continue;
                struct stat st;
                const std::string p = dir + "/" + n;
                // offsets and sizes are u32 on this wire (VeloxServer.cpp:125-146): a file beyond 4 GiB cannot be
                // served whole and would be LISTed with a wrapped size, so it is not hosted at all
                if (stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode) && (uint64_t)st.st_size <= 0xFFFFFFFFull) tracks_.push_back({n, p, (uint32_t)st.st_size});
            }
            closedir(d);
        }
        std::sort(tracks_.begin(), tracks_.end(), [](const Hosted &a, const Hosted &b) { return a.name < b.name; });
        listen_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
        if (listen_fd_ < 0) return false;
        int one = 1; setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{}; a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = htons((uint16_t)port);
        socklen_t al = sizeof(a);
        if (::bind(listen_fd_, (sockaddr *)&a, sizeof(a)) != 0 || ::listen(listen_fd_, 64) != 0 ||
            getsockname(listen_fd_, (sockaddr *)&a, &al) != 0) { ::close(listen_fd_); listen_fd_ = -1; return false; }
        port_ = ntohs(a.sin_port);
        accept_thread_ = std::thread([this] { accept_loop(); });
        return true;
    }
    int port() const { return port_; }
    size_t track_count() const { return tracks_.size(); }
    void stop()
    {
        if (stop_.exchange(true)) return;
        if (listen_fd_ >= 0) { ::shutdown(listen_fd_, SHUT_RDWR); ::close(listen_fd_); }
        if (accept_thread_.joinable()) accept_thread_.join();
        std::vector<std::unique_ptr<Client>> rest;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto &cl : clients_) if (cl->fd >= 0) ::shutdown(cl->fd, SHUT_RDWR);   // only sockets still open here
            rest.swap(clients_);
        }
        for (auto &cl : rest) if (cl->th.joinable()) cl->th.join();       // outside mu_: serve() takes it on its way out
    }
};

class RangeClient {
    int fd_ = -1;
    // One request, one framed answer.  `max_len` = the most bytes this answer may carry: a peer that
    // announces more is broken or hostile (the frame length is a u32 it controls — up to 4 GiB of
    // allocation, and more bytes than the caller made room for), so the connection is dropped instead.
    bool request(const std::string &cmd, std::vector<uint8_t> &out, uint32_t max_len)
    {
        out.clear();
        if (fd_ < 0 || !detail::send_all(fd_, cmd.data(), cmd.size())) return false;
        uint32_t be = 0;
        if (!detail::recv_all(fd_, &be, 4)) return false;
        const uint32_t n = ntohl(be);
        if (n > max_len) { ::close(fd_); fd_ = -1; return false; }           // over-answer: the stream is out of sync from here on
        out.resize(n);
        return out.empty() || detail::recv_all(fd_, out.data(), out.size());
    }

public:
    static constexpr uint32_t kMaxListBytes = 1u << 24;                       // a LIST answer: 16 MiB of "<id>|<name>|<size>" lines at most
    RangeClient() = default;
    RangeClient(const RangeClient &) = delete;
    ~RangeClient() { if (fd_ >= 0) ::close(fd_); }
    bool connect_to(const std::string &ip, int port)
    {
        fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd_ < 0) return false;
        sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
        if (inet_pton(AF_INET, ip.c_str(), &a.sin_addr) != 1 || ::connect(fd_, (sockaddr *)&a, sizeof(a)) != 0) { ::close(fd_); fd_ = -1; return false; }
        int one = 1; setsockopt(fd_, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        return true;
    }
    bool list(std::vector<TrackInfo> &out)
    {
        std::vector<uint8_t> body;
        if (!request("LIST", body, kMaxListBytes)) return false;
        out.clear();
        const std::string s(body.begin(), body.end());
        size_t p = 0;
        while (p < s.size()) {
            size_t e = s.find('\n', p); if (e == std::string::npos) e = s.size();
            const std::string line = s.substr(p, e - p);
            const size_t a = line.find('|'), b = line.rfind('|');
            if (a != std::string::npos && b != a)
                out.push_back({atoi(line.substr(0, a).c_str()), line.substr(a + 1, b - a - 1), (uint32_t)strtoul(line.substr(b + 1).c_str(), nullptr, 10)});
            p = e + 1;
        }
        return true;
    }
    // At most `length` bytes come back (the server clips the range to the file, VeloxServer.cpp:140-146);
    // an answer longer than what was asked for is refused.
    bool get(int id, uint32_t offset, uint32_t length, std::vector<uint8_t> &out)
    {
        return request("GET " + std::to_string(id) + " " + std::to_string(offset) + " " + std::to_string(length), out, length);
    }
};

// The container prologue in front of the payload (VeloxArch.h:12-26, main.cpp:185-204): 40-byte packed
// header, u32 metadata size + metadata, header blob, footer blob.
struct ContainerInfo {
    uint32_t sample_rate = 0; uint16_t channels = 0, bits = 0, format_code = 0;
    uint64_t total_samples = 0; size_t payload_offset = 0; bool ok = false;
    static uint32_t u32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
    // `have` = bytes available at p; fails (ok = false) until the whole prologue's sizes can be read.
    static ContainerInfo parse(const uint8_t *p, size_t have)
    {
        ContainerInfo c;
        if (have < 44 || u32(p) != 0x584C4556u) return c;
        const uint16_t version = (uint16_t)(p[4] | (p[5] << 8));
        c.sample_rate = u32(p + 6); c.channels = (uint16_t)(p[10] | (p[11] << 8));
        c.bits = (uint16_t)((p[12] | (p[13] << 8)) & 0x7FFF); c.format_code = (uint16_t)(p[14] | (p[15] << 8));
        c.total_samples = (uint64_t)u32(p + 16) | ((uint64_t)u32(p + 20) << 32);
        const uint32_t hb = u32(p + 24), fb = u32(p + 28);
        size_t pos = 40;
        if (version >= 0x0400) pos += 4 + (size_t)u32(p + 40);              // VeloxStreamClient.cpp:367-372
        c.payload_offset = pos + hb + fb;
        c.ok = true;
        return c;
    }
};

class Session {
    RangeClient &net_;
    int id_; uint32_t file_size_;
    std::vector<uint8_t> file_;                 // whole-file buffer, valid up to downloaded_ (VeloxStreamClient.cpp:480-481)
    size_t downloaded_ = 0;
    ContainerInfo info_;
    std::unique_ptr<VeloxCodec::StreamingDecoder> dec_;
    size_t delivered_ = 0;
    bool ended_ = false;                        // the decoder stopped before total_samples (zero chunk size, VeloxCore.h:341)
    friend class Scheduler;

public:
    static constexpr uint32_t kRange = 262144;  // bytes per GET (VeloxStreamClient.cpp:499)
    static constexpr size_t kBatch = 4096;      // samples per delivery (VeloxStreamClient.cpp:450)

    Session(RangeClient &net, const TrackInfo &t) : net_(net), id_(t.id), file_size_(t.size), file_(t.size) {}
    const ContainerInfo &info() const { return info_; }
    size_t downloaded() const { return downloaded_; }
    size_t delivered() const { return delivered_; }
    bool finished() const { return ended_ || (info_.ok && delivered_ >= info_.total_samples); }
    bool download_complete() const { return downloaded_ >= file_size_; }

    // Fetch the next range (one GET).  Returns false on a network error or when nothing is left.
    bool download_some()
    {
        if (downloaded_ >= file_size_) return false;
        std::vector<uint8_t> part;
        const uint32_t want = (uint32_t)std::min<size_t>(kRange, file_size_ - downloaded_);
        if (!net_.get(id_, (uint32_t)downloaded_, want, part) || part.empty()) return false;
        if (part.size() > file_size_ - downloaded_) return false;           // never write past the buffer the decoder borrows
        memcpy(file_.data() + downloaded_, part.data(), part.size());
        downloaded_ += part.size();
        if (!info_.ok) info_ = ContainerInfo::parse(file_.data(), downloaded_);
        // like the reference (VeloxStreamClient.cpp:336) the decoder is opened once 64 KiB or the whole
        // file are there: its constructor reads the stream header (and a float track's exponent runs)
        if (info_.ok && downloaded_ > info_.payload_offset && (downloaded_ >= 65536 || downloaded_ >= file_size_)) {
            const size_t have = downloaded_ - info_.payload_offset;
            if (!dec_) {
                // the decoder borrows the (never reallocated) file buffer; it starts out knowing only the
                // bytes that have arrived and learns about more through Extend
                dec_.reset(new VeloxCodec::StreamingDecoder(file_.data() + info_.payload_offset, have, info_.total_samples));
                // a true-float track's exponent runs may reach beyond what has arrived: try again later
                if (!dec_->HeaderComplete() && downloaded_ < file_size_) dec_.reset();
            } else dec_->Extend(have);
        }
        return true;
    }
    // Enough bytes for the decoder's next step?  The reference waits for 16 KiB beyond a linear estimate
    // of the position (VeloxStreamClient.cpp:429-437); here the decoder is asked exactly how far the
    // buffer must reach (BytesNeeded), so a slow link can never make it decode bytes that are not there.
    bool ready() const
    {
        if (!dec_ || finished()) return false;
        if (download_complete()) return true;
        return downloaded_ >= info_.payload_offset + dec_->BytesNeeded(kBatch);   // a batch may cross into the next chunk
    }
    VeloxCodec::StreamingDecoder *decoder() { return dec_.get(); }
    // Next batch of samples; 0 at the end of the track.
    size_t decode_batch(velox_sample_t *vals, uint8_t *exps, size_t cap = kBatch)
    {
        if (!dec_) return 0;
        const size_t n = dec_->DecodeMany(vals, exps, std::min(cap, (size_t)(info_.total_samples - delivered_)));
        delivered_ += n;
        if (n == 0 && download_complete()) ended_ = true;
        return n;
    }
    // Jump to `sample` (must already be downloaded far enough: the walk reads every chunk size before it).
    bool seek(size_t sample)
    {
        if (!dec_ || !dec_->Seek(sample)) return false;
        delivered_ = std::min<size_t>(sample, info_.total_samples);
        ended_ = false;
        return true;
    }
};

class Scheduler {
public:
    using Sink = std::function<void(Session &, const velox_sample_t *, const uint8_t *, size_t)>;
    // One tick: each unfinished session downloads one range if it is not ready, then all ready
    // sessions refill in ONE GPU batch and deliver one batch of samples each.  Returns the number of
    // sessions that delivered.
    size_t tick(std::vector<Session *> &sessions, const Sink &sink)
    {
        std::vector<Session *> ready;
        for (Session *s : sessions) {
            if (s->finished()) continue;
            if (!s->ready() && !s->download_complete()) s->download_some();
            if (s->ready()) ready.push_back(s);
        }
        if (ready.empty()) return 0;
        std::vector<VeloxCodec::StreamingDecoder *> decs;
        for (Session *s : ready) decs.push_back(s->decoder());
        VeloxCodec::StreamingDecoder::RefillMany(decs.data(), decs.size());
        vals_.resize(Session::kBatch); exps_.resize(Session::kBatch);
        size_t delivered = 0;
        for (Session *s : ready) {
            const size_t n = s->decode_batch(vals_.data(), exps_.data(), Session::kBatch);
            if (n) { sink(*s, vals_.data(), exps_.data(), n); delivered++; }
        }
        return delivered;
    }

private:
    std::vector<velox_sample_t> vals_; std::vector<uint8_t> exps_;
};

} // namespace velox_stream
#endif // VELOX_STREAM_POSIX_H
