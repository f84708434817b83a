// rt_kernels_frame.h — the first and the last kernels of a batch: camera rays and path state
// (k_generate), per-pixel sample sums (k_resolve), and the display pass (k_display).
#pragma once

#include "rt_wavefront.h"

using namespace rtd;

namespace {

// ---- k_generate -----------------------------------------------------------------------
// Camera ray of fragment main() (src/RayTracer.cpp:559-578 / 1333-1366): jitter
// in units of (1/W)*right and (1/H)*up, first draw horizontal.
//
// Hero-wavelength mode (src/RayTracer.cpp:1357-1364) traces four wavelengths per camera sample one
// after the other along the same camera ray, each from the same seed: the batch runs the bounce
// loop four times, and this kernel starts pass `heroPass` (0..3; -1 = mode off).  Passes after the
// first fold the finished wavelength into the sample's XYZ sum (kept in neeC, which the Spectral
// integrator does not otherwise use) and restart the path.
// get_hero_wavelengths, src/RayTracer.cpp:1259-1272, one component; mod(x, y) = x - y*floor(x/y).
__device__ __forceinline__ float heroWavelength(int k, float baseOffset) {
    const float stride = kWlRange / 4.0f;
    const float x = (kWlMin + stride * ((float)k + baseOffset)) - kWlMin;
    return (x - kWlRange * floor_(x / kWlRange)) + kWlMin;
}

template <bool SPECTRAL>
__global__ __launch_bounds__(kBlock) void k_generate(FrameParams fp, Batch bt, PathBuffers pb, int heroPass) {
    const uint32_t total = bt.K * fp.npt;
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t base = blockIdx.x * kBlock + (threadIdx.x & ~63u); base < total; base += stride) {
        uint32_t p = base + laneId();
        bool valid = false;
        RayRec r;
        float4 first = make_float4(1.0f, 1.0f, 0.0f, 0.0f);   // Spectral: the path's initial stA (throughput, previous pdf, wavelength)
        if (p < total) {
            uint32_t k, q;
            pathOf(fp, bt, p, k, q);
            int x, y;
            valid = pathPixel(fp, q, x, y);
            if (valid) {
                uint32_t g = bt.g0 + k;
                uint32_t frame = bt.firstFrame + (g / fp.S) * bt.frameStride;
                uint32_t s = g % fp.S;
                int i = (int)(s / (uint32_t)fp.ssq), j = (int)(s % (uint32_t)fp.ssq);
                uint32_t seed = hash_seed((uint32_t)x, (uint32_t)y, frame, s);

                float ux = (float)x * fp.rImgX * 2.0f - 1.0f;
                float uy = (float)y * fp.rImgY * 2.0f - 1.0f;
                V3 uv = fp.halfWidth * ux * fp.right + fp.halfHeight * uy * fp.up + fp.focalLength * fp.forward + fp.position;
                float jx = rand_float(seed);
                float jy = rand_float(seed);
                V3 d = uv + (((float)j + jx) * fp.rssq) * fp.rImgX * fp.right + (((float)i + jy) * fp.rssq) * fp.rImgY * fp.up;
                if (!SPECTRAL && !fp.stratified) d = uv + jx * fp.rImgX * fp.right + jy * fp.rImgY * fp.up;   // :589
                d = normalize(d - fp.position);

                // Throughput and previous pdf start at one and the radiance at zero, which the shading kernels know
                // at stage 0: only the wavelength is stored (Spectral), and the path's random state travels in the
                // unused word of its ray record.
                if (SPECTRAL && heroPass >= 0) {
                    // The seed is passed to traceColorWavelength by value (src/RayTracer.cpp:1051), so
                    // every wavelength of the sample starts from the same state: the one after this draw.
                    const float baseOffset = (rand_float(seed) + (float)(i * fp.ssq) + (float)j) * fp.wlDt;
                    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (heroPass > 0) {
                        sum = pb.neeC[p];
                        const float4 done = pb.radiance[p];   // the wavelength just finished: radiance, lambda
                        const V3 xyz = v3(sum.x, sum.y, sum.z) + cieXYZ(done.y) * (done.x / fp.wlPdf);
                        sum = make_float4(xyz.x, xyz.y, xyz.z, 0.0f);
                    }
                    pb.neeC[p] = sum;
                    first.z = heroWavelength(heroPass, baseOffset);
                } else if (SPECTRAL) {
                    float lambda = (rand_float(seed) + (float)(i * fp.ssq) + (float)j) * fp.wlDt + kWlMin;
                    first.z = lambda;
                }
                r.ox = fp.position.x; r.oy = fp.position.y; r.oz = fp.position.z; r.pad0 = __uint_as_float(seed);
                r.dx = d.x; r.dy = d.y; r.dz = d.z; r.pad1 = kNeverStop;
            }
        }
        const uint32_t sub = blockIdx.x % kSubQueues;
        const uint32_t at = wavePush(valid, pb.qExt[0] + (size_t)sub * pb.subCap, &pb.stages[0].ext[sub].count, p);
        if (valid) {
            const uint32_t rec = extRecord(pb, 0, sub, at);   // the camera ray sits where its queue entry does
            pb.rays[rec] = r;
            if (SPECTRAL) pb.stA[rec] = first;
        }
    }
}


// Hero-wavelength mode: the camera rays' hit records of a sample's first wavelength are kept under the path's
// number (`save`), and handed to the other three wavelengths' camera rays, whose queue positions differ (`!save`).
__global__ __launch_bounds__(kBlock) void k_camera_hits(PathBuffers pb, int save) {
    const uint32_t sub = blockIdx.x % kSubQueues;
    const uint32_t n = pb.stages[0].ext[sub].count;
    const uint32_t *q = pb.qExt[0] + (size_t)sub * pb.subCap;
    const uint32_t stride = (gridDim.x / kSubQueues) * kBlock;
    for (uint32_t i = (blockIdx.x / kSubQueues) * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t path = q[i], rec = extRecord(pb, 0, sub, i);
        if (save) pb.cameraHits[path] = pb.hits[rec];
        else pb.hits[rec] = pb.cameraHits[path];
    }
}

// ---- k_resolve ---------------------------------------------------------------------------
// Adds the batch's samples into the frame sum in sample order and, when a frame
// completes, scales by 1/ssq^2 (Spectral: XYZ -> linear sRGB) and adds it to the
// image sum (src/RayTracer.cpp:579-583, 1368-1378).  `partial` carries an
// unfinished frame's sample sum across batches.
template <bool SPECTRAL>
__global__ __launch_bounds__(kBlock) void k_resolve(FrameParams fp, Batch bt, PathBuffers pb, float4 *accum, float4 *partial) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < fp.npt; q += stride) {
        int x, y;
        if (!pathPixel(fp, q, x, y)) continue;
        const size_t px = (size_t)y * (size_t)fp.W + (size_t)x;
        V3 c = v3s(0.0f);
        float4 acc = accum[px];
        bool accDirty = false;
        for (uint32_t k = 0; k < bt.K; ++k) {
            const uint32_t s = (bt.g0 + k) % fp.S;
            if (s == 0u) {
                c = v3s(0.0f);
            } else if (k == 0u) {
                float4 pc = partial[px];
                c = v3(pc.x, pc.y, pc.z);
            }
            const uint32_t path = pathIndex(fp, bt, k, q);
            const float4 sB = pb.radiance[path];
            if (SPECTRAL) {
                const float lambda = sB.y;
                if (fp.hero) {   // hero_wavelengths_to_rgb, src/RayTracer.cpp:1274-1280: the fourth term and the mean
                    const float4 sum = pb.neeC[path];
                    c = c + (v3(sum.x, sum.y, sum.z) + cieXYZ(lambda) * (sB.x / fp.wlPdf)) / 4.0f;
                } else {
                    c = c + cieXYZ(lambda) * (sB.x / fp.wlPdf);
                }
            } else {
                c = c + v3(sB.x, sB.y, sB.z);
            }
            if (s == fp.S - 1u) {
                V3 col = (!SPECTRAL && !fp.stratified) ? c / (float)fp.rayPerPixel : c * (fp.rssq * fp.rssq);   // :592 / :582
                if (SPECTRAL) {
                    col = col / kCieYIntegral;
                    col = xyzEToLinearSRGB(col);
                }
                acc.x += col.x;
                acc.y += col.y;
                acc.z += col.z;
                acc.w += 1.0f;
                accDirty = true;
            } else if (k == bt.K - 1u) {
                partial[px] = make_float4(c.x, c.y, c.z, 0.0f);
            }
        }
        if (accDirty) accum[px] = acc;
    }
}

// ---- k_display ---------------------------------------------------------------------------
// The reference's screen pass (src/main.cpp:36-69) on the progressive mean:
// clamp below at 0, ACES filmic curve, NaN / negative / infinite debug colours,
// gamma 1/2.2; written as 8-bit RGBA (round to nearest).
__global__ __launch_bounds__(kBlock) void k_display(const float4 *accum, uchar4 *out, uint32_t pixels) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < pixels; i += gridDim.x * kBlock) {
        const float4 a = accum[i];
        const float n = a.w > 0.0f ? a.w : 1.0f;
        V3 c = vmax(v3(a.x / n, a.y / n, a.z / n), v3s(0.0f));
        c = (c * (2.51f * c + v3s(0.03f))) / (c * (2.43f * c + v3s(0.59f)) + v3s(0.14f));
        c = vclamp(c, 0.0f, 1.0f);
        V3 o;
        if (c.x != c.x || c.y != c.y || c.z != c.z) o = v3(1.0f, 0.0f, 0.0f);
        else if (c.x < 0.0f || c.y < 0.0f || c.z < 0.0f) o = v3(0.0f, 1.0f, 0.0f);
        else if (abs_(c.x) == inf() || abs_(c.y) == inf() || abs_(c.z) == inf()) o = v3(0.0f, 0.0f, 1.0f);
        else o = v3(pow_(c.x, 1.0f / 2.2f), pow_(c.y, 1.0f / 2.2f), pow_(c.z, 1.0f / 2.2f));
        out[i] = make_uchar4((unsigned char)(o.x * 255.0f + 0.5f), (unsigned char)(o.y * 255.0f + 0.5f),
                             (unsigned char)(o.z * 255.0f + 0.5f), 255);
    }
}

// ---- k_eval_colour -----------------------------------------------------------------------
// Parity hook: the Spectral integrator's colour functions and tables as the kernels see them,
// evaluated item by item.  fn: 0 cieXYZ(lambda); 1 smitsBasis(lambda, which); 2 reflectanceAt(lambda,
// rgb); 3 xyzEToLinearSRGB(xyz); 4 the tables (kCieX, kCieY, kCieZ, kSmits: 193 values per item).
__global__ __launch_bounds__(kBlock) void k_eval_colour(int fn, const float *in, uint32_t items, float *out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < items; i += gridDim.x * kBlock) {
        if (fn == 0) {
            const V3 r = cieXYZ(in[i]);
            out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
        } else if (fn == 1) {
            out[i] = smitsBasis(in[2 * i], (int)in[2 * i + 1]);
        } else if (fn == 2) {
            out[i] = reflectanceAt(in[4 * i], v3(in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]));
        } else if (fn == 3) {
            const V3 r = xyzEToLinearSRGB(v3(in[3 * i], in[3 * i + 1], in[3 * i + 2]));
            out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
        } else {
            for (int k = 0; k < 41; ++k) { out[193 * i + k] = kCieX[k]; out[193 * i + 41 + k] = kCieY[k]; out[193 * i + 82 + k] = kCieZ[k]; }
            for (int k = 0; k < 70; ++k) out[193 * i + 123 + k] = kSmits[k];
        }
    }
}

}  // namespace
