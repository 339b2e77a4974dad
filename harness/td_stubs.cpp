// td_stubs.cpp -- the build's own stub layer for running the reference's UNMODIFIED top/td.cpp (its four threads, rings and semaphores,
// td.cpp:82-223, 306-855) against the drop-in tracker library, SURVEY 8(f)#3.  Everything here is written from scratch against the
// PUBLIC declarations td.cpp compiles against -- the vendored OpenCV 3.3 headers (videoio.hpp: cv::VideoCapture; mat.hpp: cv::Mat;
// cvstd.hpp: cv::String; highgui.hpp: imshow / waitKey) and td.cpp:17-41 (the detector DLL's tensor* entry points) -- and replaces exactly
// what the demo takes from OpenCV's libraries and from yolo3.dll:
//   * the camera: frames of a synthetic scene read from the file MOT_HARNESS_INPUT (written by the test from
//     multiple_object_tracking_amd/synth.py) instead of a capture device;
//   * the detector: the boxes stored with each frame instead of a YOLOv3 network;
//   * the display: imshow() only counts frames.
// The capture stub hands out frame f + 1 only after frame f has reached imshow(): the demo then always runs with batches of one frame,
// which sidesteps the semaphore accounting of td.cpp:196-223 (it posts sem_imgs_prc once per BATCH but waits once per FRAME).
// After the last frame "HARNESS_DONE" is printed; a newline on stdin ends main() (td.cpp:842).
// These stubs pin nothing about the oracle: they only let the reference's real tracker thread drive the library.
//
// input file: int32 nframes; then per frame: 1280 * 720 * 3 BGR bytes, int32 nbox, nbox * bbox_t (24 bytes each, top/cnntype.h:36-41)
#include "opencv2/imgproc/imgproc.hpp"
#include "opencv2/highgui/highgui.hpp"
#include "cnntype.h"

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <chrono>
#include <vector>

namespace {

const size_t kFrameBytes = (size_t)MTCNN_IMGW * MTCNN_IMGH * 3;

struct Input {
    int nframes = 0;
    std::vector<std::vector<unsigned char>> frames;
    std::vector<std::vector<bbox_t>> boxes;
    bool load()
    {
        const char* path = std::getenv("MOT_HARNESS_INPUT");
        if (!path) { std::fprintf(stderr, "td_stubs: MOT_HARNESS_INPUT is not set\n"); return false; }
        FILE* f = std::fopen(path, "rb");
        if (!f) { std::fprintf(stderr, "td_stubs: cannot open %s\n", path); return false; }
        int n = 0;
        if (std::fread(&n, sizeof n, 1, f) != 1 || n < 1 || n > 4096) { std::fclose(f); return false; }
        frames.resize(n); boxes.resize(n);
        for (int i = 0; i < n; i++) {
            frames[i].resize(kFrameBytes);
            int nb = 0;
            if (std::fread(frames[i].data(), 1, kFrameBytes, f) != kFrameBytes || std::fread(&nb, sizeof nb, 1, f) != 1 || nb < 0 || nb > 128) { std::fclose(f); return false; }
            boxes[i].resize(nb);
            if (nb && std::fread(boxes[i].data(), sizeof(bbox_t), nb, f) != (size_t)nb) { std::fclose(f); return false; }
        }
        std::fclose(f);
        nframes = n;
        return true;
    }
};

Input g_in;
bool g_open = false;
std::mutex g_mu;
std::condition_variable g_cv;
int g_served = 0, g_shown = 0;                 // frames handed to the capture thread / frames that reached imshow
// a few frame buffers in rotation (at most MAX_GPU_BATCH = 4 frames are in flight, td.cpp:764); which frame a buffer holds
const int kRing = 8;
unsigned char* g_buf[kRing]; int g_buf_frame[kRing];

} // namespace

namespace cv {

// ---- cv::VideoCapture (videoio.hpp): every virtual has to exist for the class's vtable ----
VideoCapture::VideoCapture() {}
VideoCapture::VideoCapture(const String&) {}
VideoCapture::VideoCapture(const String&, int) {}
VideoCapture::VideoCapture(int) {}
VideoCapture::~VideoCapture() {}
bool VideoCapture::open(const String&) { return false; }
bool VideoCapture::open(const String&, int) { return false; }
bool VideoCapture::open(int)
{
    g_open = g_in.load();
    for (int i = 0; i < kRing; i++) { g_buf[i] = (unsigned char*)std::malloc(kFrameBytes); g_buf_frame[i] = -1; }
    return g_open;
}
bool VideoCapture::isOpened() const { return g_open; }
void VideoCapture::release() {}
bool VideoCapture::grab() { return g_open; }
bool VideoCapture::retrieve(OutputArray, int) { return false; }
bool VideoCapture::read(OutputArray) { return false; }
bool VideoCapture::set(int, double) { return true; }
double VideoCapture::get(int) const { return 0.0; }
VideoCapture& VideoCapture::operator>>(UMat&) { return *this; }

VideoCapture& VideoCapture::operator>>(Mat& m)
{
    int f;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        g_cv.wait(lk, [] { return g_shown == g_served; });             // the previous frame has been through the whole pipeline
        f = g_served;
        if (f >= g_in.nframes) {
            lk.unlock();
            std::printf("HARNESS_DONE\n"); std::fflush(stdout);
            for (;;) std::this_thread::sleep_for(std::chrono::seconds(3600));   // main() ends the process on its newline
        }
        g_served++;
    }
    unsigned char* buf = g_buf[f % kRing];
    std::memcpy(buf, g_in.frames[f].data(), kFrameBytes);
    g_buf_frame[f % kRing] = f;
    // a header over caller-owned memory (no UMatData: release() frees nothing), the public fields of cv::Mat (mat.hpp)
    m.flags = Mat::MAGIC_VAL | CV_8UC3 | Mat::CONTINUOUS_FLAG;
    m.dims = 2; m.rows = MTCNN_IMGH; m.cols = MTCNN_IMGW;
    m.data = buf; m.datastart = buf; m.dataend = buf + kFrameBytes; m.datalimit = buf + kFrameBytes;
    m.allocator = 0; m.u = 0;
    m.step.p[0] = (size_t)MTCNN_IMGW * 3; m.step.p[1] = 3;
    return *this;
}

// ---- cv::Mat / cv::String out-of-line members the headers' inline code refers to ----
void Mat::deallocate() {}                                              // (never reached: u == 0)

char* String::allocate(size_t len)
{
    // cvstd.hpp's contract: a reference count in the int in front of the characters
    const size_t total = ((len + 1 + sizeof(int) - 1) / sizeof(int)) * sizeof(int);
    int* data = (int*)std::malloc(total + sizeof(int));
    data[0] = 1;
    cstr_ = (char*)(data + 1);
    len_ = len;
    cstr_[len] = 0;
    return cstr_;
}

void String::deallocate()
{
    int* data = (int*)cstr_;
    len_ = 0; cstr_ = 0;
    if (data && __sync_fetch_and_add(data - 1, -1) == 1) std::free(data - 1);
}

// ---- highgui ----
void imshow(const String&, InputArray)
{
    { std::lock_guard<std::mutex> lk(g_mu); g_shown++; }
    g_cv.notify_all();
}
int waitKey(int) { return -1; }

} // namespace cv

// ---- the detector DLL (td.cpp:17-41): boxes of the synthetic scene instead of a network ----
extern "C" int tensorStartup(const char*, const char*, const char* [3]) { return 0; }
extern "C" int tensorCleanup() { return 0; }
extern "C" int tensorRunB(int batchSize, int, int, int*, int*, unsigned char** pimgbuf, bbox_chain_t** pbbox, yolo3_options_t*)
{
    for (int i = 0; i < batchSize; i++) {
        int f = -1;
        for (int k = 0; k < kRing; k++) if (g_buf[k] == pimgbuf[i]) f = g_buf_frame[k];
        pbbox[i]->nbox = 0;
        if (f < 0) continue;
        const std::vector<bbox_t>& b = g_in.boxes[f];
        pbbox[i]->nbox = (int)b.size();
        for (size_t j = 0; j < b.size(); j++) pbbox[i]->bbox[j] = b[j];
    }
    return 0;
}
