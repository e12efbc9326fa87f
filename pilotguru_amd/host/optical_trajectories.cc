// optical_trajectories.cc -- the drop-in CLI surface of pilotguru's optical_trajectories
// (src/optical_trajectories.cc:36-114) over libpgorb, FRONT-END MODE.
//
// Same flags as the reference (vocabulary_file, camera_settings, out_dir, in_video, visualize,
// vertical_flip, horizontal_flip, output_per_segment_videos, rotation_smooth_sigma; gflags
// syntax --flag=value / --flag value / --noflag) and the same three CHECKs (:77-79).  What runs
// per frame is the part of the reference this project rebuilds: ORB extraction
// (Frame::ExtractORB), Frame::ComputeBoW, the 64x48 grid and ORBmatcher(0.9,true)
// .SearchForInitialization(previous, current, ..., 100) (Tracking.cc:596-597).  The SLAM back
// end (Tracking/LocalMapping/LoopClosing, pose estimation) is out of scope, so instead of
// trajectory-<k>.json (which needs poses) the run writes <out_dir>/frontend-<k>.json; the
// trajectory JSON writer is exercised with --trajectory_in=<poses.txt> (see trajectory_json.hpp),
// (the frame loop streams batches through pgorb_stream_*: page-locked input slots, upload / kernels / download overlapped)
// and --poses_in=<poses.txt> runs the whole tail of TrackImageSequence on poses from any back end
// (src/slam/track_image_sequence.cc:63-109: heading smoothing, PCA plane, eigenvalue gate,
// projected directions, turn angles, trajectory-<segment_id>.json).
// --devices=a,b,...: one ride over several GPUs from ONE process: one host thread per device, the
// vocabulary parsed once and broadcast with one RCCL collective behind the C ABI (pgorb_comm_create_local + pgorb_vocab_broadcast;
// the reference shares one ORBVocabulary by pointer, src/optical_trajectories.cc:87-94), shard r of the frames on the r-th device,
// the report merged in rank order: frontend-0.json and the --dump_features file equal the single-device run's byte for byte.
// --shard=rank/world (with --device): one ride over several processes, one per GPU -- SURVEY.md section 8(e): contiguous
// chunks of frames with a one-frame overlap, nothing exchanged; process `rank` writes frontend-<rank>.json (and its
// --dump_features file) for the frames it owns, the same rule as pilotguru_amd/dist.py frame_chunk_for_rank.
// No libav here: --in_video takes a .y4m (Y plane), a printf pattern of PGM (grey) or PPM (RGB24) files
// (frames/%06d.pgm), or a headerless .gray / .rgb (interleaved RGB24, what the reference's reader decodes to:
// src/io/image_sequence_reader.cc:138-208) file sized by Camera_width / Camera_height.  Frames go into the stream's
// page-locked slots exactly as read; --vertical_flip / --horizontal_flip (:53-58), --rotation=0|90|180|270 (the
// reader takes it from the container's metadata, :186-205) and Tracking's cvtColor (Tracking.cc:247-260, channel
// order by Camera_RGB) run on the device in front of the pyramid (pgorb_stream_create_ingest).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "orb_extractor.hpp"
#include "trajectory_json.hpp"
#include "trajectory_post.hpp"

namespace {

struct Flags {
    std::string vocabulary_file, camera_settings, out_dir, in_video, trajectory_in, poses_in, dump_features;
    bool visualize = true, vertical_flip = false, horizontal_flip = false, output_per_segment_videos = false;
    bool init_extractor = false;       // front-end mode has no map: --init_extractor treats the ride as "not initialised yet"
    bool vocabulary_cache = true;      // <vocabulary_file>.pgvoc beside the text file (--novocabulary_cache: always parse the text)
    long long rotation_smooth_sigma = -1;
    int device = 0, batch = 64, max_frames = -1, segment_id = 0, rotation = 0;
    int copy_threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));     // slot filling and the per-frame BoW maps
    int shard_rank = 0, shard_world = 1;          // --shard=rank/world: this process takes its chunk of the ride
    std::vector<int> devices;                     // --devices=a,b,...: one process, one host thread per device, shard r on device r of the list
};

[[noreturn]] void check_failed(const char* what)
{
    fprintf(stderr, "Check failed: %s\n", what);             // glog CHECK aborts; we exit non-zero
    exit(EXIT_FAILURE);
}

bool parse_flags(int argc, char** argv, Flags& F)
{
    std::map<std::string, std::string*> str = {{"vocabulary_file", &F.vocabulary_file}, {"camera_settings", &F.camera_settings},
        {"out_dir", &F.out_dir}, {"in_video", &F.in_video}, {"trajectory_in", &F.trajectory_in}, {"poses_in", &F.poses_in}, {"dump_features", &F.dump_features}};
    std::map<std::string, bool*> bl = {{"visualize", &F.visualize}, {"vertical_flip", &F.vertical_flip},
        {"horizontal_flip", &F.horizontal_flip}, {"output_per_segment_videos", &F.output_per_segment_videos},
        {"init_extractor", &F.init_extractor}, {"vocabulary_cache", &F.vocabulary_cache}};
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--", 0) == 0) a = a.substr(2); else if (a.rfind("-", 0) == 0) a = a.substr(1); else return false;
        std::string name = a, val; bool has = false;
        const size_t eq = a.find('=');
        if (eq != std::string::npos) { name = a.substr(0, eq); val = a.substr(eq + 1); has = true; }
        if (bl.count(name)) { *bl[name] = !has || (val != "false" && val != "0" && val != "no"); continue; }
        if (name.rfind("no", 0) == 0 && bl.count(name.substr(2))) { *bl[name.substr(2)] = false; continue; }
        if (!has) { if (i + 1 >= argc) return false; val = argv[++i]; }
        if (str.count(name)) *str[name] = val;
        else if (name == "rotation_smooth_sigma") F.rotation_smooth_sigma = atoll(val.c_str());
        else if (name == "device") F.device = atoi(val.c_str());
        else if (name == "batch") F.batch = atoi(val.c_str());
        else if (name == "max_frames") F.max_frames = atoi(val.c_str());
        else if (name == "segment_id") F.segment_id = atoi(val.c_str());
        else if (name == "copy_threads") F.copy_threads = atoi(val.c_str());
        else if (name == "rotation") {
            F.rotation = atoi(val.c_str());
            if (F.rotation != 0 && F.rotation != 90 && F.rotation != 180 && F.rotation != 270) {
                fprintf(stderr, "ERROR: unsupported rotation %d: only multiples of 90 degrees\n", F.rotation); return false; }   // reader :203-207
        }
        else if (name == "shard") {
            if (sscanf(val.c_str(), "%d/%d", &F.shard_rank, &F.shard_world) != 2 || F.shard_world < 1 || F.shard_rank < 0 ||
                F.shard_rank >= F.shard_world) { fprintf(stderr, "ERROR: --shard wants rank/world with 0 <= rank < world\n"); return false; }
        }
        else if (name == "devices") {
            F.devices.clear();
            std::stringstream ss(val); std::string tok;
            while (std::getline(ss, tok, ',')) {
                char* e = nullptr; const long d = strtol(tok.c_str(), &e, 10);
                if (tok.empty() || *e || d < 0 || d > 1023) { fprintf(stderr, "ERROR: --devices wants a comma-separated list of device ordinals\n"); return false; }
                F.devices.push_back((int)d);
            }
            if (F.devices.empty()) { fprintf(stderr, "ERROR: --devices wants at least one device\n"); return false; }
        }
        else { fprintf(stderr, "ERROR: unknown command line flag '%s'\n", name.c_str()); return false; }
    }
    if (!F.devices.empty() && F.shard_world > 1) { fprintf(stderr, "ERROR: --devices (one process, every shard) and --shard (one process per shard) exclude each other\n"); return false; }
    return true;
}

// OpenCV FileStorage YAML as written by calibrate.cc:504-544: "key: value" lines
std::map<std::string, double> read_settings(const std::string& path)
{
    std::map<std::string, double> m;
    std::ifstream f(path);
    if (!f.good()) check_failed("camera_settings file is readable");
    std::string line;
    while (std::getline(f, line)) {
        const size_t c = line.find(':');
        if (c == std::string::npos || line[0] == '%' || line[0] == '-') continue;
        std::string k = line.substr(0, c), v = line.substr(c + 1);
        k.erase(0, k.find_first_not_of(" \t")); k.erase(k.find_last_not_of(" \t") + 1);
        char* e = nullptr;
        const double d = strtod(v.c_str(), &e);
        if (e != v.c_str()) m[k] = d;
    }
    return m;
}

struct FrameSource {                  // ImageSequenceSource (include/io/image_sequence_reader.hpp:23-28)
    std::string path; int w = 0, h = 0; double fps = 30; long frame = 0;
    int channels = 1;                 // 1 = grey plane, 3 = interleaved RGB24 (.rgb file, PPM sequence)
    bool y4m = false, pattern = false; size_t y4mFrameBytes = 0;
    // .y4m / .gray files are mapped: a frame goes from the page cache into the page-locked slot with ONE user-space
    // copy (read() straight into page-locked memory measured 25 % slower than read() + memcpy, the mapping is faster than both)
    const uint8_t* map = nullptr; size_t mapSize = 0, pos = 0;
    const uint8_t* lastPtr = nullptr;     // where the frame next() just stepped over starts in the mapping
    std::string patHead, patTail; int patWidth = 0; bool patZero = false;      // "<head>%0Nd<tail>", parsed once
    // The user's path is never handed to printf as a format: exactly one %d / %Nd / %0Nd conversion is
    // accepted ("%%" is a literal per cent sign) and the frame name is assembled by hand.
    bool parse_pattern(const std::string& p)
    {
        bool seen = false;
        std::string* cur = &patHead;
        for (size_t i = 0; i < p.size(); i++) {
            if (p[i] != '%') { *cur += p[i]; continue; }
            if (i + 1 < p.size() && p[i + 1] == '%') { *cur += '%'; i++; continue; }
            if (seen) return false;
            size_t j = i + 1;
            if (j < p.size() && p[j] == '0') { patZero = true; j++; }
            int wd = 0;
            while (j < p.size() && p[j] >= '0' && p[j] <= '9' && wd < 100) wd = wd * 10 + (p[j++] - '0');
            if (j >= p.size() || p[j] != 'd' || wd > 32) return false;
            patWidth = wd; seen = true; cur = &patTail; i = j;
        }
        return seen;
    }
    std::string frame_name(long k) const
    {
        std::string num = std::to_string(k);
        if ((int)num.size() < patWidth) num.insert(0, (size_t)patWidth - num.size(), patZero ? '0' : ' ');
        return patHead + num + patTail;
    }
    bool open(const std::string& p, int sw, int sh, double f)
    {
        path = p; fps = f;
        if (p.find('%') != std::string::npos) { pattern = true; return parse_pattern(p); }
        const int fd = ::open(p.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || sb.st_size <= 0) { ::close(fd); return false; }
        mapSize = (size_t)sb.st_size;
        void* m = mmap(nullptr, mapSize, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) { map = nullptr; return false; }
        map = static_cast<const uint8_t*>(m);
        (void)madvise(m, mapSize, MADV_SEQUENTIAL);
        if (p.size() > 4 && p.substr(p.size() - 4) == ".y4m") {
            y4m = true;
            const void* nl = memchr(map, '\n', std::min<size_t>(mapSize, 511));
            if (!nl) return false;
            char hdr[512]; const size_t hl = (size_t)((const uint8_t*)nl - map);
            memcpy(hdr, map, hl); hdr[hl] = 0; pos = hl + 1;
            int chroma = 420; char* t = strtok(hdr, " \n");
            while (t) {
                if (t[0] == 'W') w = atoi(t + 1); else if (t[0] == 'H') h = atoi(t + 1);
                else if (t[0] == 'F') { int a = 30, b = 1; if (sscanf(t + 1, "%d:%d", &a, &b) == 2 && b) fps = (double)a / b; }
                else if (t[0] == 'C') chroma = !strncmp(t + 1, "mono", 4) ? 400 : !strncmp(t + 1, "444", 3) ? 444 : !strncmp(t + 1, "422", 3) ? 422 : 420;
                t = strtok(nullptr, " \n");
            }
            const size_t y = (size_t)w * h;
            y4mFrameBytes = chroma == 400 ? y : chroma == 444 ? 3 * y : chroma == 422 ? 2 * y : y + 2 * ((size_t)((w + 1) / 2) * ((h + 1) / 2));
            return w > 0 && h > 0;
        }
        w = sw; h = sh;
        if (p.size() > 4 && p.substr(p.size() - 4) == ".rgb") channels = 3;
        return w > 0 && h > 0;
    }
    size_t frame_bytes() const { return (size_t)w * h * channels; }
    ~FrameSource() { if (map) munmap(const_cast<uint8_t*>(map), mapSize); }
    // number of frames of the sequence (pattern: probe for the first missing file)
    long count()
    {
        if (pattern) {
            long k = 0;
            for (;; k++) { FILE* f = fopen(frame_name(k).c_str(), "rb"); if (!f) break; fclose(f); }
            return k;
        }
        if (y4m) return (long)((mapSize - pos) / (y4mFrameBytes + 6));         // "FRAME\n" + planes (frame headers without parameters)
        return (long)(mapSize / frame_bytes());
    }
    // start at frame k (frame ids and timestamps stay those of the whole sequence)
    bool skip(long k)
    {
        if (pattern) { frame = k; return true; }
        std::vector<uint8_t> g; long long t, id;
        while (frame < k) if (!next(g, &t, &id, nullptr, false)) return false;      // (walks the frame headers, copies nothing)
        return true;
    }
    // The next frame's grey plane goes to `dst` (w * h bytes: the page-locked slot of the stream, no staging copy) or,
    // when dst is null (the first frame: its size is not known yet for a PGM sequence), into `gray`.
    bool next(std::vector<uint8_t>& gray, long long* time_usec, long long* frame_id, uint8_t* dst = nullptr, bool copy = true)
    {
        if (pattern) {
            FILE* f = fopen(frame_name(frame).c_str(), "rb");
            if (!f) return false;
            char magic[3] = {0}; int maxv = 0, fw = 0, fh = 0;
            if (fscanf(f, "%2s %d %d %d", magic, &fw, &fh, &maxv) != 4 || (strcmp(magic, "P5") && strcmp(magic, "P6")) || maxv != 255) { fclose(f); return false; }
            const int fch = magic[1] == '6' ? 3 : 1;
            if (frame > 0 && w > 0 && fch != channels) { fclose(f); fprintf(stderr, "ERROR: frame %ld changes the pixel format\n", frame); return false; }
            channels = fch;
            // every frame of a sequence has the first frame's size (the context and the buffers are sized once)
            if (fw <= 0 || fh <= 0 || (w > 0 && (fw != w || fh != h))) {
                fclose(f);
                fprintf(stderr, "ERROR: frame %ld is %dx%d, the sequence started with %dx%d\n", frame, fw, fh, w, h);
                return false;
            }
            w = fw; h = fh;
            fgetc(f);
            if (!dst) { gray.resize(frame_bytes()); dst = gray.data(); }
            const bool ok = fread(dst, 1, frame_bytes(), f) == frame_bytes();
            fclose(f);
            if (!ok) return false;
        } else {
            const size_t y = frame_bytes();
            size_t skip = 0;
            if (y4m) {                                                 // "FRAME[ params]\n" then the planes; only Y is used
                if (pos + 5 > mapSize || memcmp(map + pos, "FRAME", 5)) return false;
                const void* nl = memchr(map + pos, '\n', std::min<size_t>(mapSize - pos, 63));
                if (!nl) return false;
                pos = (size_t)((const uint8_t*)nl - map) + 1;
                skip = y4mFrameBytes - y;
            }
            if (pos + y + skip > mapSize) return false;
            lastPtr = map + pos;
            if (copy) {
                if (!dst) { gray.resize(y); dst = gray.data(); }
                memcpy(dst, map + pos, y);
            }
            pos += y + skip;
        }
        *frame_id = frame;
        *time_usec = (long long)llround((double)frame * 1e6 / fps);
        frame++;
        return true;
    }
};

int write_trajectory_from_text(const Flags& F)
{
    // fixture format: first line 6 plane doubles; then per point:
    // time_usec is_lost frame_id tx ty tz qw qx qy qz dir_x dir_y turn_angle
    std::ifstream f(F.trajectory_in);
    if (!f.good()) check_failed("trajectory_in file is readable");
    // numbers go through strtod so that "nan" / "inf" can be fed to the writer (operator>> refuses them)
    auto num = [&](double* v) { std::string tok; if (!(f >> tok)) return false; char* e = nullptr; *v = strtod(tok.c_str(), &e); return e != tok.c_str(); };
    double plane[6];
    for (double& v : plane) num(&v);
    std::vector<pgorb::PoseWithTimestamp> traj; std::vector<double> dirs, turns;
    for (;;) {
        pgorb::PoseWithTimestamp p; long long t, id; int lost; double dx, dy, turn;
        if (!(f >> t >> lost >> id) || !num(&p.pose.translation[0]) || !num(&p.pose.translation[1]) || !num(&p.pose.translation[2]) ||
            !num(&p.pose.qw) || !num(&p.pose.qx) || !num(&p.pose.qy) || !num(&p.pose.qz) || !num(&dx) || !num(&dy) || !num(&turn)) break;
        p.time_usec = t; p.is_lost = lost != 0; p.frame_id = id;
        traj.push_back(p); dirs.push_back(dx); dirs.push_back(dy); turns.push_back(turn);
    }
    const std::string out = F.out_dir + "/trajectory-0.json";            // TrajectoryOutFileName (:65-70)
    std::ofstream o(out);
    if (!o.good()) check_failed("out_dir is writable");
    o << pgorb::trajectory_to_json(plane, traj, dirs.data(), turns.data(), 0) << std::endl;   // dump(2) << endl (:108-109)
    return EXIT_SUCCESS;
}

// The tail of TrackImageSequence (src/slam/track_image_sequence.cc:63-109) on poses read from text:
// one "time_usec is_lost frame_id tx ty tz qw qx qy qz" line per trajectory point.
int write_trajectory_from_poses(const Flags& F)
{
    std::ifstream f(F.poses_in);
    if (!f.good()) check_failed("poses_in file is readable");
    std::vector<pgorb::PoseWithTimestamp> trajectory;
    for (;;) {
        pgorb::PoseWithTimestamp p; long long t, id; int lost;
        if (!(f >> t >> lost >> id >> p.pose.translation[0] >> p.pose.translation[1] >> p.pose.translation[2] >> p.pose.qw >>
              p.pose.qx >> p.pose.qy >> p.pose.qz)) break;
        p.time_usec = t; p.is_lost = lost != 0; p.frame_id = id;
        trajectory.push_back(p);
    }
    if (trajectory.empty()) { fprintf(stderr, "empty trajectory, nothing written\n"); return EXIT_SUCCESS; }       // :64-66
    if (F.rotation_smooth_sigma > 0 && !pgorb::SmoothHeadingDirections(&trajectory, (int)F.rotation_smooth_sigma))  // :68-70
        check_failed("SmoothHeadingDirections");
    pgorb::TrajectoryPCA pca;
    if (!pgorb::TrajectoryToPCA(trajectory, &pca)) check_failed("trajectory has at least 3 poses (cv::PCA over 3 x N)");
    if (pca.eigenvalues[2] > pca.eigenvalues[1] * 1e-2) {                                                           // :83-90
        fprintf(stderr, "3rd eigenvalue was too large, dropping the trajectory. Relative magnitude wrt the 2nd eigenvalue: %g\n",
                pca.eigenvalues[2] / pca.eigenvalues[1]);
        return EXIT_SUCCESS;
    }
    const double* plane = pca.eigenvectors;                                                                          // rowRange(0, 2), :92
    const std::vector<double> dirs = pgorb::ProjectDirections(trajectory, plane);
    const std::vector<double> turns = pgorb::Projected2DDirectionsToTurnAngles(dirs);
    const std::string out = F.out_dir + "/trajectory-" + std::to_string(F.segment_id) + ".json";
    std::ofstream o(out);
    if (!o.good()) check_failed("out_dir is writable");
    o << pgorb::trajectory_to_json(plane, trajectory, dirs.data(), turns.data(), 0) << std::endl;
    return EXIT_SUCCESS;
}

// Extractor parameters of the ride (underscore keys of this fork, Tracking.cc:131-135; defaults as written by calibrate.cc:518-532)
struct RideParams {
    int nFeatures = 2000, nLevels = 8, iniTh = 20, minTh = 7, camW = 0, camH = 0, rgb = 0;
    float scaleFactor = 1.2f;
    double fps = 30.0;
};

// One shard of the ride on one device: the whole single-device run is the shard 0 / 1.  With --shard=r/N the process runs
// shard r; with --devices=a,b,... the process runs every shard, one host thread per device.
// Frames [firstExtracted, stop) are read, frames from firstOwned on are reported (the frame before firstOwned is extracted
// only as the predecessor of the first owned match) -- the same rule as pilotguru_amd/dist.py frame_chunk_for_rank.
struct Shard {
    const Flags& F;
    const RideParams& R;
    int device, rank, world;
    FrameSource src;
    pgorb::ORBextractor* ext = nullptr;
    pgorb_stream* st = nullptr;
    std::string devError;
    double tCtx = 0, tStream = 0;
    long firstOwned = 0, maxFrames = -1, total = 0;
    std::vector<uint8_t> frame0;                              // the first frame tells the size
    long long t0 = 0, id0 = 0;
    bool frame0Read = false, haveSize = false;
    int upW = 0, upH = 0;                                     // the upright frame the extractor sees
    std::string frames;                                       // the "frames" entries of the report, comma separated
    std::string dumpBuf;                                      // --dump_features records when several shards share one file
    FILE* dump = nullptr;                                     //   ... or the file itself (one shard)
    double loopSec = 0;

    Shard(const Flags& f, const RideParams& r, int dev, int rk, int wd) : F(f), R(r), device(dev), rank(rk), world(wd) {}

    void open_source()
    {
        if (!src.open(F.in_video, R.camW, R.camH, R.fps))
            check_failed("input video opens (y4m, PGM pattern or .gray + Camera_width/Camera_height)");
        maxFrames = F.max_frames;
        if (world > 1) {
            long nframes = src.count();
            if (F.max_frames >= 0) nframes = std::min<long>(nframes, F.max_frames);
            const long per = nframes / world, extra = nframes % world;
            const long first = rank * per + std::min<long>(rank, extra), stop = first + per + (rank < extra ? 1 : 0);
            const long firstExtracted = stop > first ? std::max<long>(first - 1, 0) : first;
            firstOwned = first;
            if (!src.skip(firstExtracted)) check_failed("input video holds the frames of this shard");
            maxFrames = stop - firstExtracted;
        }
        // (a printf pattern of PGM / PPM files only knows its frame size once the first file is read: read it first then)
        if (src.w <= 0 || src.h <= 0) {
            if (!((maxFrames < 0 || maxFrames > 0) && src.next(frame0, &t0, &id0))) frame0.clear();
            frame0Read = true;
        }
        const bool swapSides = F.rotation == 90 || F.rotation == 270;        // the reader's rotation swaps the sides
        upW = swapSides ? src.h : src.w; upH = swapSides ? src.w : src.h;
        haveSize = src.w > 0 && src.h > 0;                    // (false: an empty pattern source / an empty shard of one -- nothing to extract)
    }

    // HIP runtime start, the context's arenas, the stream's page-locked slots (beside the vocabulary load)
    void create_device(const std::function<double()>& since)
    {
        if (!haveSize) return;
        const int B = std::max(1, F.batch), DEPTH = 3;
        try { ext = new pgorb::ORBextractor(R.nFeatures, R.scaleFactor, R.nLevels, R.iniTh, R.minTh, upW, upH, B, device); }
        catch (const std::exception& e) { devError = e.what(); return; }
        tCtx = since();
        if (pgorb_max_keypoints(ext->context(), upW, upH) < 0) { devError = "frame size usable for the ORB cell grid"; return; }
        // frames as read; rotation, flips and the grey conversion on the device (Camera_RGB: 1 = RGB, 0 = BGR; Tracking.cc:247-260).
        // A settings file without the key means BGR: `int nRGB = fSettings["Camera_RGB"]` reads 0 from an empty cv::FileNode (Tracking.cc:102)
        if (pgorb_stream_create_ingest(ext->context(), src.w, src.h, src.channels, R.rgb != 0, F.rotation,
                                       F.vertical_flip, F.horizontal_flip, B, DEPTH, &st) != PGORB_OK) { devError = pgorb_last_error(ext->context()); return; }
        tStream = since();
    }

    // The frame loop of TrackImageSequence (src/slam/track_image_sequence.cc:43-52) as a stream of batches
    // (include/pgorb.h, pgorb_stream_*): the source writes every frame straight into a page-locked slot; upload,
    // kernels and result download of up to DEPTH batches overlap.  The per-frame work of the tracking thread --
    // Frame::ComputeBoW's transform and MonocularInitialization's SearchForInitialization(previous, current) -- runs on
    // the device for the whole batch as the stream's front-end stage (pgorb_stream_frontend); the host only folds the
    // per-feature words into BowVector / FeatureVector and writes the report.  The vocabulary is resident in the context.
    void run(int vs, int vwt, int copyThreads)
    {
        const int B = std::max(1, F.batch), DEPTH = 3;
        if (!frame0Read && !((maxFrames < 0 || maxFrames > 0) && src.next(frame0, &t0, &id0))) frame0.clear();
        if (ext) {
            // Frame::ComputeImageBounds without distortion: [0, cols] x [0, rows] (Frame.cc:462-466); ORBmatcher(0.9, true)
            // .SearchForInitialization(mInitialFrame, mCurrentFrame, mvbPrevMatched, mvIniMatches, 100) (Tracking.cc:596-597);
            // transform(..., 4) (Frame.cc:404)
            if (pgorb_stream_frontend(st, 0.f, (float)upW, 0.f, (float)upH, 100, 0.9f, 1, 4) != PGORB_OK) check_failed(pgorb_last_error(ext->context()));
        }
        const size_t fbytes = src.frame_bytes();
        std::vector<std::vector<long long>> tusS(DEPTH, std::vector<long long>(B)), idsS(DEPTH, std::vector<long long>(B));
        std::ostringstream js;
        long read = 0; bool first = true, firstOfRide = true;
        int submitted = 0, collected = 0; bool more = !frame0.empty();
        const auto loopStart = std::chrono::steady_clock::now();
        std::vector<int> nbowB, nfvB;
        while (more || collected < submitted) {
            // keep DEPTH batches in flight: fill and submit the next slot while there are frames
            while (more && submitted - collected < DEPTH) {
                const int slot = submitted % DEPTH;
                uint8_t* in = pgorb_stream_input(st, slot);
                int nb = 0;
                std::vector<uint8_t> tmp;
                // a mapped file's frames are copied into the page-locked slot by several threads at once (one thread moves
                // ~6-9 GB/s out of the page cache: 3 000 grey / 1 400 RGB24 1080p frames per second, a third of the link)
                std::vector<std::pair<const uint8_t*, uint8_t*>> copies;
                while (nb < B && (maxFrames < 0 || read < maxFrames)) {
                    if (read == 0) { memcpy(in, frame0.data(), fbytes); tusS[slot][0] = t0; idsS[slot][0] = id0; }
                    else if (src.map) {
                        if (!src.next(tmp, &tusS[slot][nb], &idsS[slot][nb], nullptr, false)) { more = false; break; }
                        copies.emplace_back(src.lastPtr, in + (size_t)nb * fbytes);
                    } else {
                        if (!src.next(tmp, &tusS[slot][nb], &idsS[slot][nb], in + (size_t)nb * fbytes)) { more = false; break; }
                    }
                    nb++; read++;
                }
                if (!copies.empty()) {
                    const int nthreads = (int)std::min<size_t>(copies.size(), (size_t)std::max(1, copyThreads));
                    std::vector<std::thread> pool;
                    for (int t = 1; t < nthreads; t++)
                        pool.emplace_back([&, t] { for (size_t k = t; k < copies.size(); k += nthreads) memcpy(copies[k].second, copies[k].first, fbytes); });
                    for (size_t k = 0; k < copies.size(); k += nthreads) memcpy(copies[k].second, copies[k].first, fbytes);
                    for (auto& th : pool) th.join();
                }
                if (maxFrames >= 0 && read >= maxFrames) more = false;
                if (!nb) break;
                if (pgorb_stream_submit(st, slot, nb) != PGORB_OK) check_failed(pgorb_last_error(ext->context()));
                submitted++;
            }
            if (collected >= submitted) break;
            const int slot = collected % DEPTH;
            const int32_t* n = nullptr; const pgorb_keypoint* kps = nullptr; const uint8_t* desc = nullptr; int cap = 0;
            const int nb = pgorb_stream_wait(st, slot, &n, &kps, &desc, nullptr, nullptr, nullptr, &cap);
            if (nb < 0) check_failed(pgorb_last_error(ext->context()));
            const int32_t* nmatch = nullptr; const uint32_t *word = nullptr, *node = nullptr; const double* wt = nullptr;
            if (pgorb_stream_frontend_results(st, slot, nullptr, &nmatch, &word, &wt, &node) != PGORB_OK) check_failed(pgorb_last_error(ext->context()));
            collected++;
            const std::vector<long long>&tus = tusS[slot], &ids = idsS[slot];
            // Frame::ComputeBoW: BowVector / FeatureVector of transform(descriptors, ..., 4)  (Frame.cc:399-406): the per-feature words
            // come from the device, the two ordered maps are built on the host -- ~0.1 ms per frame, so the frames of a batch are
            // shared out over the copy threads (one thread held the whole loop to ~5 000 frames/s)
            nbowB.assign(nb, 0); nfvB.assign(nb, 0);
            {
                const int nthreads = std::max(1, std::min(nb, copyThreads));
                auto work = [&](int t) {
                    std::vector<uint32_t> bid, fnode, ffeat; std::vector<double> bval; std::vector<int32_t> fstart;
                    for (int i = t; i < nb; i += nthreads) {
                        if (ids[i] < firstOwned || !n[i]) continue;
                        const size_t o = (size_t)i * cap;
                        bid.resize(n[i] + 1); bval.resize(n[i] + 1); fnode.resize(n[i] + 1); ffeat.resize(n[i] + 1); fstart.resize(n[i] + 2);
                        pgorb_bow_vectors(n[i], word + o, wt + o, node + o, vs, vwt, bid.data(), bval.data(), &nbowB[i],
                                          fnode.data(), fstart.data(), ffeat.data(), &nfvB[i]);
                    }
                };
                std::vector<std::thread> pool;
                for (int t = 1; t < nthreads; t++) pool.emplace_back(work, t);
                work(0);
                for (auto& th : pool) th.join();
            }
            for (int i = 0; i < nb; i++) {
                const bool hadPrev = !firstOfRide;
                firstOfRide = false;
                if (ids[i] < firstOwned) continue;               // the overlap frame of a shard: only a predecessor
                const size_t o = (size_t)i * cap;
                const int nbow = nbowB[i], nfv = nfvB[i];
                const int nmatches = hadPrev ? nmatch[i] : -1;   // MonocularInitialization's matcher call (Tracking.cc:596-597)
                js << (first ? "" : ",\n") << "    {\"frame_id\": " << ids[i] << ", \"time_usec\": " << tus[i] << ", \"n_keypoints\": " << n[i]
                   << ", \"n_bow_words\": " << nbow << ", \"n_feature_nodes\": " << nfv << ", \"n_matches_prev\": " << nmatches << "}";
                first = false;
                if (dump || !F.dump_features.empty()) {
                    const int32_t hdr[2] = {(int32_t)ids[i], n[i]};
                    if (dump) {
                        fwrite(hdr, 4, 2, dump);
                        fwrite(kps + o, sizeof(pgorb_keypoint), n[i], dump);
                        fwrite(desc + o * 32, 32, n[i], dump);
                    } else {
                        dumpBuf.append((const char*)hdr, 8);
                        dumpBuf.append((const char*)(kps + o), sizeof(pgorb_keypoint) * (size_t)n[i]);
                        dumpBuf.append((const char*)(desc + o * 32), (size_t)32 * n[i]);
                    }
                }
            }
            for (int i = 0; i < nb; i++) total += ids[i] >= firstOwned;      // (a shard's overlap frame is not one of its frames)
        }
        frames = js.str();
        loopSec = std::chrono::duration<double>(std::chrono::steady_clock::now() - loopStart).count();
    }
};

}  // namespace

int main(int argc, char** argv)
{
    Flags F;
    if (!parse_flags(argc, argv, F)) return EXIT_FAILURE;
    if (!F.trajectory_in.empty()) return write_trajectory_from_text(F);
    if (!F.poses_in.empty()) return write_trajectory_from_poses(F);
    if (F.vocabulary_file.empty()) check_failed("!FLAGS_vocabulary_file.empty()");     // :77
    if (F.camera_settings.empty()) check_failed("!FLAGS_camera_settings.empty()");     // :78
    if (F.in_video.empty()) check_failed("!FLAGS_in_video.empty()");                   // :79

    std::map<std::string, double> S = read_settings(F.camera_settings);
    auto get = [&](const char* k, double d) { return S.count(k) ? S[k] : d; };
    RideParams R;
    // mpIniORBextractor: 2 * nFeatures while the tracker is NOT_INITIALIZED / NO_IMAGES_YET (Tracking.cc:137-143, :262-264) --
    // the frames MonocularInitialization runs SearchForInitialization on (:596-597)
    R.nFeatures = (int)get("ORBextractor_nFeatures", 2000) * (F.init_extractor ? 2 : 1); R.nLevels = (int)get("ORBextractor_nLevels", 8);
    R.scaleFactor = (float)get("ORBextractor_scaleFactor", 1.2);
    R.iniTh = (int)get("ORBextractor_iniThFAST", 20); R.minTh = (int)get("ORBextractor_minThFAST", 7);
    R.camW = (int)get("Camera_width", 0); R.camH = (int)get("Camera_height", 0); R.fps = get("Camera_fps", 30.0);
    R.rgb = (int)get("Camera_RGB", 0);

    // Start-up (round 4): creating the device context -- HIP runtime start, the context's arenas, the stream's page-locked slots --
    // was 0.45 s of a 0.56-s run on a 2 048-frame clip, one thing after the other.  Now a second thread (one per device) creates
    // context and stream while this one loads the vocabulary (from its binary cache beside the text file when there is one).
    const auto tStart = std::chrono::steady_clock::now();
    std::function<double()> since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count(); };
    const bool timing = getenv("PGORB_CLI_TIMING") != nullptr;
    auto epoch = [] { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); };
    if (timing) fprintf(stderr, "main entered at epoch %.6f\n", epoch());

    // --devices=a,b,...: ONE process, one host thread per listed device, shard r of the ride on device r of the list; the
    // vocabulary is parsed ONCE and reaches every device with one RCCL broadcast (pgorb_vocab_broadcast); the shards' entries
    // are merged in rank order, so frontend-0.json and the --dump_features file equal the single-device run's byte for byte.
    // --shard=r/N (one process per GPU, started by a launcher): this process is shard r and writes frontend-<r>.json.
    const bool multi = !F.devices.empty();
    std::vector<Shard*> shards;
    if (multi) for (size_t r = 0; r < F.devices.size(); r++) shards.push_back(new Shard(F, R, F.devices[r], (int)r, (int)F.devices.size()));
    else shards.push_back(new Shard(F, R, F.device, F.shard_rank, F.shard_world));
    for (Shard* sh : shards) sh->open_source();
    std::vector<std::thread> devThreads;
    for (Shard* sh : shards) devThreads.emplace_back([sh, &since] { sh->create_device(since); });

    pgorb_vocab* voc = nullptr;
    int vocFromCache = 0;
    if ((F.vocabulary_cache ? pgorb_vocab_load_cached(F.vocabulary_file.c_str(), &voc, &vocFromCache)
                            : pgorb_vocab_load_text(F.vocabulary_file.c_str(), &voc)) != PGORB_OK) {           // ORBVocabulary.cc:8 CHECK
        for (auto& t : devThreads) t.join();
        check_failed("vocabulary loads (ORB vocabulary text file)");
    }
    int vk, vL, vn, vw, vs, vwt; pgorb_vocab_info(voc, &vk, &vL, &vn, &vw, &vs, &vwt);
    const double tVoc = since();
    for (auto& t : devThreads) t.join();
    for (Shard* sh : shards) if (!sh->devError.empty()) check_failed(sh->devError.c_str());

    // one vocabulary for every System (src/optical_trajectories.cc:87-94): a plain upload for one context, ONE broadcast for several
    double tBcast = 0;
    if (!multi) { if (shards[0]->ext && pgorb_vocab_upload(shards[0]->ext->context(), voc) != PGORB_OK) check_failed("vocabulary upload"); }
    else {
        std::vector<pgorb_ctx*> ctxs;
        for (Shard* sh : shards) if (sh->ext) ctxs.push_back(sh->ext->context());
        if (!ctxs.empty()) {
            pgorb_comm* comm = nullptr;
            if (pgorb_comm_create_local(ctxs.data(), (int)ctxs.size(), &comm) != PGORB_OK) check_failed(pgorb_last_error(ctxs[0]));
            if (pgorb_vocab_broadcast(comm, 0, voc, &tBcast) != PGORB_OK) check_failed(pgorb_last_error(ctxs[0]));
            fprintf(stderr, "vocabulary: %d nodes parsed once, broadcast to %d context(s) on %d device(s) over RCCL in %.3f ms\n",
                    vn, (int)ctxs.size(), pgorb_comm_ranks(comm), tBcast * 1e3);
            pgorb_comm_destroy(comm);
        }
    }
    if (timing)
        fprintf(stderr, "start-up: vocabulary %s at %.3f s, context at %.3f s, stream at %.3f s, vocabulary resident at %.3f s\n",
                vocFromCache ? "(from its cache)" : "(text parsed)", tVoc, shards[0]->tCtx, shards[0]->tStream, since());

    FILE* dump = F.dump_features.empty() ? nullptr : fopen(F.dump_features.c_str(), "wb");
    if (!multi) shards[0]->dump = dump;
    const auto loopStart = std::chrono::steady_clock::now();
    {
        // the copy threads (slot filling, the per-frame BoW maps) are shared out over the device threads
        const int perShard = std::max(1, F.copy_threads / (int)shards.size());
        std::vector<std::thread> run;
        for (size_t r = 1; r < shards.size(); r++) run.emplace_back([&, r] { shards[r]->run(vs, vwt, perShard); });
        shards[0]->run(vs, vwt, perShard);
        for (auto& t : run) t.join();
    }
    long total = 0;
    std::ostringstream js;
    js << "{\n  \"frames\": [";
    bool first = true;
    for (Shard* sh : shards) {
        total += sh->total;
        if (sh->frames.empty()) continue;
        js << (first ? "\n" : ",\n") << sh->frames;
        first = false;
        if (multi && dump) fwrite(sh->dumpBuf.data(), 1, sh->dumpBuf.size(), dump);
    }
    js << "\n  ],\n  \"orb\": {\"nFeatures\": " << R.nFeatures << ", \"nLevels\": " << R.nLevels << ", \"iniThFAST\": " << R.iniTh
       << ", \"minThFAST\": " << R.minTh << "},\n  \"vocabulary\": {\"k\": " << vk << ", \"L\": " << vL << ", \"nodes\": " << vn << ", \"words\": " << vw
       << "}\n}";
    if (dump) fclose(dump);
    const std::string out = (F.out_dir.empty() ? std::string(".") : F.out_dir) + "/frontend-" + std::to_string(!multi && F.shard_world > 1 ? F.shard_rank : 0) + ".json";
    {
        std::ofstream o(out);
        if (!o.good()) check_failed("out_dir is writable");
        o << js.str() << std::endl;
    }
    const double loopSec = std::chrono::duration<double>(std::chrono::steady_clock::now() - loopStart).count();
    fprintf(stderr, "optical_trajectories (front-end mode): %ld frames -> %s (frame loop: %.3f s, %.0f frames/s%s)\n", total, out.c_str(),
            loopSec, loopSec > 0 ? total / loopSec : 0.0, multi ? (", " + std::to_string(shards.size()) + " device threads").c_str() : "");
    if (timing) fprintf(stderr, "report written at %.3f s, epoch %.6f\n", since(), epoch());
    if (getenv("PGORB_CLI_TEARDOWN")) {                       // orderly teardown (leak checkers): unpinning and freeing the arenas takes ~0.2 s
        for (Shard* sh : shards) {
            if (sh->st) pgorb_stream_destroy(sh->st);
            delete sh->ext;
            delete sh;
        }
        pgorb_vocab_free(voc);
        return EXIT_SUCCESS;
    }
    // the report is on disk (closed above): leave without unmapping gigabytes of page-locked and device memory one piece at a time
    fflush(nullptr);
    _exit(EXIT_SUCCESS);
}
