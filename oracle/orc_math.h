// TEST INFRASTRUCTURE - CPU oracle, not part of the shipped renderer.
//
// f32 vector / matrix arithmetic with the operation order of the crates the reference
// computes with: nalgebra 0.33.2 / nalgebra-glm 0.19.0 (Cargo.lock:661-678).  Their source is
// not part of the reference tree, so these are restatements of the published algorithms; the
// call sites they serve are listed next to each function.  Must be compiled with
// -ffp-contract=off and without -ffast-math: every + - * / below is one IEEE-754 binary32
// operation, exactly as rustc emits them.
#pragma once
#include <cmath>
#include <cstdint>

namespace orc {

struct Vec3 { float x, y, z; };
struct Vec4 { float x, y, z, w; };

inline Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 v3(const float* p) { return Vec3{p[0], p[1], p[2]}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// Matrix::component_mul (raytracer.rs:148-150)
inline Vec3 cmul(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }

// glm::dot for Vec3 - nalgebra's unrolled 3-vector dot: (a0*b0 + a1*b1) + a2*b2.
// Call sites: intersect.rs:47-51,63-64,120,126,150,157,161,164,369; raytracer.rs:144-145,194,204,230-231.
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// glm::cross - intersect.rs:149,160; triangle.rs:43.
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Matrix::norm = sqrt(norm_squared), norm_squared = dot(v, v).
inline float norm(Vec3 a) { return std::sqrt(dot(a, a)); }

// glm::normalize / Matrix::normalize: unscale by the norm, a true division per component and
// no zero guard (a zero vector becomes NaN).  raytracer.rs:45,88,105,107,167,219;
// intersect.rs:60,170,305,313; plane.rs:18; sphere.rs:14; triangle.rs:43.
inline Vec3 normalize(Vec3 a) { float n = norm(a); return {a.x / n, a.y / n, a.z / n}; }

// glm::distance(p0, p1) = norm(p1 - p0).  raytracer.rs:178; intersect.rs:318.
inline float distance(Vec3 p0, Vec3 p1) { return norm(p1 - p0); }

// f32::min / f32::max: if one operand is NaN the other is returned (IEEE minNum/maxNum).
// raytracer.rs:144-145; intersect.rs:94,207-208; hull.rs:25-31.
inline float fmin32(float a, float b) { return std::fmin(a, b); }
inline float fmax32(float a, float b) { return std::fmax(a, b); }

// Column-major 4x4, m[4*col + row], the storage order of glm::Mat4.
struct Mat4 { float m[16]; };

inline Mat4 mat_identity() {
    Mat4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}

// Matrix * vector as nalgebra's gemv computes it for statically sized operands: the result
// starts as column0 * x0 and every further column is added with axcpy, i.e.
// r_i = ((m_i0*x0 + m_i1*x1) + m_i2*x2) + m_i3*x3.  raytracer.rs:38-39; intersect.rs:298.
inline Vec4 mul(const Mat4& a, Vec4 v) {
    float r[4];
    for (int i = 0; i < 4; ++i) {
        float acc = a.m[i] * v.x;
        acc = a.m[4 + i] * v.y + acc;
        acc = a.m[8 + i] * v.z + acc;
        acc = a.m[12 + i] * v.w + acc;
        r[i] = acc;
    }
    return {r[0], r[1], r[2], r[3]};
}

// Matrix * matrix = one gemv per column of the right operand.  transform.rs:30,59,76; instance.rs:24.
inline Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int j = 0; j < 4; ++j) {
        Vec4 c = mul(a, Vec4{b.m[4 * j], b.m[4 * j + 1], b.m[4 * j + 2], b.m[4 * j + 3]});
        r.m[4 * j] = c.x; r.m[4 * j + 1] = c.y; r.m[4 * j + 2] = c.z; r.m[4 * j + 3] = c.w;
    }
    return r;
}

// glm::inverse -> Matrix::try_inverse for a 4x4: nalgebra's unrolled cofactor expansion (the
// well-known MESA gluInvertMatrix formulas on the column-major slice), determinant from the
// first row of cofactors, then every entry scaled by 1/det.  A singular matrix gives zeros
// (glm::inverse's unwrap_or_else).  transform.rs:62,77; instance.rs:25,28.
inline Mat4 inverse(const Mat4& a) {
    const float* m = a.m;
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    Mat4 r{};
    if (det == 0.0f) return r;
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float inv_det = 1.0f / det;
    for (int k = 0; k < 16; ++k) r.m[k] = inv[k] * inv_det;
    return r;
}

// f32::to_radians and glm::radians: x * (PI_f32 / 180).  transform.rs:49; parser.rs:40,222; scene.rs:61.
inline float radians(float deg) {
    const float pi = 3.14159265358979323846f;
    return deg * (pi / 180.0f);
}

}  // namespace orc
