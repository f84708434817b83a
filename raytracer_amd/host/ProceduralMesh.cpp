#include "ProceduralMesh.h"

#include <cmath>

namespace procedural {

namespace {

u32 mix32(u32 x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Appends the parallelogram q, q+u, q+u+v, q+v as two triangles (q,q+u,q+u+v) and (q,q+u+v,q+v).
void addQuad(MeshData &m, glm::vec3 q, glm::vec3 u, glm::vec3 v, i32 material, glm::vec3 normal = glm::vec3(0.0f), float uvScale = 1.0f) {
    const int base = static_cast<int>(m.vertices.size());
    const glm::vec3 p[4] = {q, q + u, q + u + v, q + v};
    const glm::vec2 t[4] = {{0, 0}, {uvScale, 0}, {uvScale, uvScale}, {0, uvScale}};
    for (int k = 0; k < 4; ++k) m.vertices.push_back({p[k], normal, t[k]});
    MeshData::Identifier a, b;
    a.index = glm::ivec3(base, base + 1, base + 2);
    b.index = glm::ivec3(base, base + 2, base + 3);
    a.materialIndex = b.materialIndex = material;
    m.identifiers.push_back(a);
    m.identifiers.push_back(b);
}

Material matte(glm::vec3 albedo) {
    Material m;
    m.albedo = albedo;
    return m;
}

}  // namespace

MeshData CornellBox(glm::vec3 pos, float L, float lightLen, float emission, bool front, glm::vec3 wallAlbedo) {
    MeshData lights, mesh;
    Material lamp;
    lamp.albedo = glm::vec3(0.0f);
    lamp.emissionColor = glm::vec3(1.0f);
    lamp.emissionStrength = emission;
    mesh.materials = {lamp, matte(glm::vec3(.65f, .05f, .05f)), matte(glm::vec3(.12f, .45f, .15f)), matte(wallAlbedo)};

    const glm::vec3 ex{L, 0, 0}, ey{0, L, 0}, ez{0, 0, L};
    addQuad(lights, pos + glm::vec3{(L - lightLen) * 0.5f, L - 1e-3f, (L - lightLen) * 0.5f}, glm::vec3{lightLen, 0, 0}, glm::vec3{0, 0, lightLen}, 0);

    addQuad(mesh, pos, ey, ez, 1);       // left, red
    addQuad(mesh, pos + ex, ey, ez, 2);  // right, green
    addQuad(mesh, pos, ex, ez, 3);       // floor
    addQuad(mesh, pos + ey, ex, ez, 3);  // ceiling
    addQuad(mesh, pos + ez, ey, ex, 3);  // back
    if (front) addQuad(mesh, pos, ex, ey, 3);

    mesh.adoptLights(lights);
    mesh.deriveTextureFlags();
    return mesh;
}

MeshData KnotTube(int nu, int nv, glm::vec3 center, float scale, u32 seed, const Material &material) {
    MeshData mesh;
    mesh.materials = {material};
    mesh.vertices.resize(static_cast<size_t>(nu) * nv);

    const double twoPi = 6.283185307179586476925286766559;
    auto curve = [&](double s, double out[3]) {
        const double r = 2.0 + std::cos(3.0 * s);
        out[0] = r * std::cos(2.0 * s);
        out[1] = r * std::sin(2.0 * s);
        out[2] = -std::sin(3.0 * s);
    };
    // A handful of seeded harmonics displace the tube radius ("scales" of the stand-in dragon).
    double amp[4], fu[4], fv[4], ph[4];
    for (int k = 0; k < 4; ++k) {
        amp[k] = 0.04 + 0.03 * (mix32(seed + 11u * k) % 1000) / 1000.0;
        fu[k] = 7 + static_cast<int>(mix32(seed + 101u * k) % 40);
        fv[k] = 2 + static_cast<int>(mix32(seed + 1009u * k) % 6);
        ph[k] = twoPi * (mix32(seed + 7919u * k) % 1000) / 1000.0;
    }

    for (int i = 0; i < nu; ++i) {
        const double s = twoPi * i / nu;
        double c[3], cn[3];
        curve(s, c);
        curve(s + 1e-4, cn);
        double T[3] = {cn[0] - c[0], cn[1] - c[1], cn[2] - c[2]};
        double tl = std::sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
        for (double &x : T) x /= tl;
        // Frame from the knot's centre direction (never parallel to the tangent for this curve).
        double R[3] = {c[0], c[1], 0.0};
        double d = R[0] * T[0] + R[1] * T[1] + R[2] * T[2];
        double N[3] = {R[0] - d * T[0], R[1] - d * T[1], R[2] - d * T[2]};
        double nl = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        for (double &x : N) x /= nl;
        double B[3] = {T[1] * N[2] - T[2] * N[1], T[2] * N[0] - T[0] * N[2], T[0] * N[1] - T[1] * N[0]};

        for (int j = 0; j < nv; ++j) {
            const double a = twoPi * j / nv;
            double rad = 0.45;
            for (int k = 0; k < 4; ++k) rad += amp[k] * std::sin(fu[k] * s + fv[k] * a + ph[k]);
            const double ca = std::cos(a), sa = std::sin(a);
            MeshData::Vertex &v = mesh.vertices[static_cast<size_t>(i) * nv + j];
            double p[3];
            for (int k = 0; k < 3; ++k) p[k] = c[k] + rad * (ca * N[k] + sa * B[k]);
            // Knot lives in [-3.6,3.6]^2 x [-1.6,1.6]; rotate so its flat side faces the camera (x,y stay, z thin).
            v.position = center + glm::vec3(static_cast<float>(p[0] * scale), static_cast<float>(p[1] * scale), static_cast<float>(p[2] * scale));
            v.normal = glm::vec3(0.0f);
            v.uv = glm::vec2(static_cast<float>(static_cast<double>(i) / nu), static_cast<float>(static_cast<double>(j) / nv));
        }
    }

    mesh.identifiers.reserve(static_cast<size_t>(nu) * nv * 2);
    for (int i = 0; i < nu; ++i) {
        const int i1 = (i + 1) % nu;
        for (int j = 0; j < nv; ++j) {
            const int j1 = (j + 1) % nv;
            const int v00 = i * nv + j, v10 = i1 * nv + j, v11 = i1 * nv + j1, v01 = i * nv + j1;
            MeshData::Identifier a, b;
            a.index = glm::ivec3(v00, v10, v11);
            b.index = glm::ivec3(v00, v11, v01);
            a.materialIndex = b.materialIndex = 0;
            mesh.identifiers.push_back(a);
            mesh.identifiers.push_back(b);
        }
    }

    // Area-weighted smooth normals, accumulated in identifier order.
    for (const auto &id : mesh.identifiers) {
        const glm::vec3 &A = mesh.vertices[id.index.x].position, &B = mesh.vertices[id.index.y].position, &C = mesh.vertices[id.index.z].position;
        const glm::vec3 fn = glm::cross(B - A, C - A);
        mesh.vertices[id.index.x].normal += fn;
        mesh.vertices[id.index.y].normal += fn;
        mesh.vertices[id.index.z].normal += fn;
    }
    for (auto &v : mesh.vertices) {
        if (glm::dot(v.normal, v.normal) > 0.0f) v.normal = glm::normalize(v.normal);
    }
    mesh.deriveTextureFlags();
    return mesh;
}

MeshData Prism(glm::vec3 center, float side, float depth, const Material &material) {
    MeshData mesh;
    mesh.materials = {material};
    const float h = side * 0.8660254f;
    const glm::vec3 a{-side * 0.5f, -h / 3.0f, 0}, b{side * 0.5f, -h / 3.0f, 0}, c{0, 2.0f * h / 3.0f, 0};
    const glm::vec3 dz{0, 0, depth * 0.5f};
    const glm::vec3 P[6] = {center + a - dz, center + b - dz, center + c - dz, center + a + dz, center + b + dz, center + c + dz};
    for (const glm::vec3 &p : P) mesh.vertices.push_back({p, glm::vec3(0.0f), glm::vec2(0.0f)});
    const int tris[8][3] = {{0, 2, 1}, {3, 4, 5}, {0, 1, 4}, {0, 4, 3}, {1, 2, 5}, {1, 5, 4}, {2, 0, 3}, {2, 3, 5}};
    for (const auto &t : tris) {
        MeshData::Identifier id;
        id.index = glm::ivec3(t[0], t[1], t[2]);
        id.materialIndex = 0;
        mesh.identifiers.push_back(id);
    }
    mesh.deriveTextureFlags();
    return mesh;
}

MeshData LightPanel(glm::vec3 q, glm::vec3 u, glm::vec3 v, glm::vec3 colour, float strength) {
    MeshData lights, mesh;
    Material lamp;
    lamp.albedo = glm::vec3(0.0f);
    lamp.emissionColor = colour;
    lamp.emissionStrength = strength;
    mesh.materials = {lamp};
    addQuad(lights, q, u, v, 0);
    mesh.adoptLights(lights);
    mesh.deriveTextureFlags();
    return mesh;
}

MeshData::Texture CheckerTexture(int width, int height, int channels, u32 seed, int wrapS, int wrapT) {
    MeshData::Texture t;
    t.width = width;
    t.height = height;
    t.channels = channels;
    t.wrapS = wrapS;
    t.wrapT = wrapT;
    t.data.resize(static_cast<size_t>(width) * height * channels);
    const int cell = (width >= 16) ? width / 8 : 1;
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const bool on = ((x / cell) + (y / cell)) & 1;
            for (int c = 0; c < channels; ++c) {
                const u32 noise = mix32(seed ^ (static_cast<u32>(y) * 73856093u) ^ (static_cast<u32>(x) * 19349663u) ^ (static_cast<u32>(c) * 83492791u)) & 63u;
                u32 level = (on ? 150u + 30u * c : 40u + 20u * c) + noise;
                if (c == 3) level = on ? 255u : (noise < 32u ? 0u : 255u);  // alpha: opaque squares, stippled gaps
                if (level > 255u) level = 255u;
                t.data[(static_cast<size_t>(y) * width + x) * channels + c] = static_cast<float>(level) / 255.0f;
            }
        }
    }
    return t;
}

MeshData TexturedRoom(glm::vec3 pos, float L, int texSize, u32 seed) {
    MeshData lights, mesh;

    Material lamp;
    lamp.albedo = glm::vec3(0.0f);
    lamp.emissionColor = glm::vec3(1.0f, 0.95f, 0.9f);
    lamp.emissionStrength = 18.0f;
    lamp.texture.emissiveTexture = 3;

    Material floorMat;  // base colour + metallic-roughness + normal map
    floorMat.albedo = glm::vec3(0.9f);
    floorMat.roughness = 0.8f;
    floorMat.metallic = 0.6f;
    floorMat.texture.baseColorTexture = 0;
    floorMat.texture.metallicRoughnessTexture = 1;
    floorMat.texture.normalTexture = 2;
    floorMat.normalScale = 0.7f;

    Material wallMat;  // base colour (3-channel) + occlusion
    wallMat.albedo = glm::vec3(0.8f);
    wallMat.texture.baseColorTexture = 4;
    wallMat.texture.occlusionTexture = 1;
    wallMat.occlusionStrength = 0.5f;

    Material card;  // alpha-cut base colour
    card.albedo = glm::vec3(0.3f, 0.7f, 0.3f);
    card.alphaCut = 0.5f;
    card.texture.baseColorTexture = 0;

    Material glass;  // transmission texture
    glass.albedo = glm::vec3(0.95f);
    glass.transmission = 0.9f;
    glass.roughness = 0.05f;
    glass.texture.transmissionTexture = 1;
    glass.texture.baseColorTexture = 4;

    mesh.materials = {lamp, floorMat, wallMat, card, glass};
    mesh.textures = {CheckerTexture(texSize, texSize, 4, seed + 1),
                     CheckerTexture(texSize, texSize, 3, seed + 2, 33071, 33648),
                     CheckerTexture(texSize, texSize, 3, seed + 3, 33648, 10497),
                     CheckerTexture(texSize / 2, texSize / 2, 3, seed + 4),
                     CheckerTexture(texSize, texSize / 2, 3, seed + 5, 10497, 33071)};

    const glm::vec3 ex{L, 0, 0}, ey{0, L, 0}, ez{0, 0, L};
    addQuad(lights, pos + glm::vec3{0.3f * L, L - 1e-3f, 0.3f * L}, glm::vec3{0.4f * L, 0, 0}, glm::vec3{0, 0, 0.4f * L}, 0, glm::vec3(0, -1, 0), 1.0f);

    addQuad(mesh, pos, ex, ez, 1, glm::vec3(0, 1, 0), 3.0f);         // floor, uv beyond [0,1] exercises wrapping
    addQuad(mesh, pos, ey, ez, 2, glm::vec3(1, 0, 0), 2.0f);         // left
    addQuad(mesh, pos + ex, ey, ez, 2, glm::vec3(-1, 0, 0), 2.0f);   // right
    addQuad(mesh, pos + ey, ex, ez, 2, glm::vec3(0, -1, 0), 1.0f);   // ceiling
    addQuad(mesh, pos + ez, ey, ex, 2, glm::vec3(0, 0, -1), 1.5f);   // back
    addQuad(mesh, pos + glm::vec3{0.2f * L, 0, 0.6f * L}, glm::vec3{0.6f * L, 0, 0}, glm::vec3{0, 0.6f * L, 0}, 3, glm::vec3(0, 0, -1), 1.0f);  // card
    addQuad(mesh, pos + glm::vec3{0.1f * L, 0.1f * L, 0.35f * L}, glm::vec3{0.35f * L, 0, 0.1f * L}, glm::vec3{0, 0.5f * L, 0}, 4, glm::vec3(0.0f), 1.0f);  // glass pane

    mesh.adoptLights(lights);
    mesh.deriveTextureFlags();
    return mesh;
}

MeshData RandomSoup(glm::vec3 pos, float L, int nTris, int nMaterials, int nLights, u32 seed) {
    u32 state = seed * 2654435761u + 12345u;
    auto u01 = [&state]() {
        state = mix32(state + 0x9E3779B9u);
        return static_cast<float>(state >> 8) * (1.0f / 16777216.0f);
    };
    auto pick = [&](std::initializer_list<float> corners, float lo, float hi) {
        // One time in three a corner value, otherwise uniform in [lo, hi].
        if (u01() < 0.34f) {
            const size_t k = static_cast<size_t>(u01() * static_cast<float>(corners.size()));
            return *(corners.begin() + (k < corners.size() ? k : corners.size() - 1));
        }
        return lo + (hi - lo) * u01();
    };

    MeshData lights, mesh;
    const int nTextures = 4;
    for (int t = 0; t < nTextures; ++t)
        mesh.textures.push_back(CheckerTexture(4 + 4 * t, 8, t % 2 ? 3 : 4, seed + 17u * static_cast<u32>(t), t == 1 ? 33071 : 10497, t == 2 ? 33648 : 10497));
    nMaterials = nMaterials < 2 ? 2 : nMaterials;
    for (int m = 0; m < nMaterials; ++m) {
        Material mat;
        mat.albedo = glm::vec3(pick({0.0f, 1.0f}, 0.05f, 0.95f), pick({0.0f, 1.0f}, 0.05f, 0.95f), pick({0.0f, 1.0f}, 0.05f, 0.95f));
        mat.subsurface = pick({0.0f, 1.0f}, 0.0f, 1.0f);
        mat.roughness = pick({0.0f, 1.0f, 1e-4f}, 0.02f, 1.0f);
        mat.metallic = pick({0.0f, 1.0f}, 0.0f, 1.0f);
        mat.specular = pick({0.0f, 1.0f, 0.5f}, 0.0f, 1.0f);
        mat.specularTint = pick({0.0f, 1.0f}, 0.0f, 1.0f);
        mat.transmission = pick({0.0f, 0.0f, 1.0f}, 0.0f, 1.0f);
        mat.ior = pick({1.0f, 1.5f, 2.4f}, 1.01f, 2.0f);
        mat.normalScale = pick({-1.0f, 1.0f}, 0.1f, 2.0f);
        mat.occlusionStrength = pick({0.0f, 1.0f}, 0.0f, 1.0f);
        if (m == 0) {                      // material 0 is the light set's
            mat.emissionColor = glm::vec3(1.0f, 0.9f, 0.8f);
            mat.emissionStrength = 12.0f;
        } else if (u01() < 0.15f) {        // stray emitters that are not in the light set
            mat.emissionColor = glm::vec3(u01(), u01(), u01());
            mat.emissionStrength = 3.0f * u01();
        }
        if (m > 0 && u01() < 0.5f) {
            auto tex = [&]() { return u01() < 0.5f ? static_cast<i32>(u01() * nTextures) % nTextures : -1; };
            mat.texture.baseColorTexture = tex();
            mat.texture.metallicRoughnessTexture = tex();
            mat.texture.normalTexture = tex();
            mat.texture.emissiveTexture = tex();
            mat.texture.occlusionTexture = tex();
            mat.texture.transmissionTexture = tex();
            if (mat.texture.baseColorTexture >= 0 && u01() < 0.5f) mat.alphaCut = 0.2f + 0.6f * u01();
        }
        mesh.materials.push_back(mat);
    }

    auto addTriangle = [&](MeshData &dst, i32 material) {
        const glm::vec3 c = pos + glm::vec3(u01(), u01(), u01()) * L;
        const float size = (u01() < 0.2f ? 0.6f : 0.15f) * L * (0.2f + u01());
        glm::vec3 p[3];
        for (auto &v : p) v = c + glm::vec3(u01() - 0.5f, u01() - 0.5f, u01() - 0.5f) * size;
        const float kind = u01();
        if (kind < 0.02f) p[2] = p[1];                                   // zero-area triangle
        else if (kind < 0.04f) p[2] = p[0] + (p[1] - p[0]) * 2.0f;        // three collinear points
        const i32 base = static_cast<i32>(dst.vertices.size());
        glm::vec3 g = glm::cross(p[1] - p[0], p[2] - p[0]);
        const float normalKind = u01();
        for (int k = 0; k < 3; ++k) {
            MeshData::Vertex v;
            v.position = p[k];
            if (normalKind < 0.3f) v.normal = glm::vec3(0.0f);                                         // geometric normal is used
            else if (normalKind < 0.7f && glm::dot(g, g) > 0.0f) v.normal = glm::normalize(g);
            else v.normal = glm::vec3(u01() - 0.5f, u01() - 0.5f, u01() - 0.5f);                      // skewed, not unit length
            v.uv = glm::vec2(u01() * 3.0f - 1.0f, u01() * 3.0f - 1.0f);                                // beyond [0,1]: wrapping
            dst.vertices.push_back(v);
        }
        MeshData::Identifier id;
        id.index = glm::ivec3(base, base + 1, base + 2);
        id.materialIndex = material;
        dst.identifiers.push_back(id);
    };
    for (int t = 0; t < nLights; ++t) addTriangle(lights, 0);
    for (int t = nLights; t < nTris; ++t) addTriangle(mesh, 1 + static_cast<i32>(u01() * static_cast<float>(nMaterials - 1)) % (nMaterials - 1));
    mesh.adoptLights(lights);
    mesh.deriveTextureFlags();
    return mesh;
}

MeshData CoplanarStack(glm::vec3 pos, float edge, int n, int copies, u32 seed) {
    MeshData mesh;
    n = n < 1 ? 1 : n;
    copies = copies < 2 ? 2 : copies;
    u32 state = seed * 2246822519u + 97u;
    auto next = [&state]() { state = mix32(state + 0x9E3779B9u); return state; };
    for (int c = 0; c < copies; ++c) {
        Material mat;
        const float h = static_cast<float>(c) / static_cast<float>(copies);
        mat.albedo = glm::vec3(0.15f + 0.8f * h, 0.9f - 0.7f * h, (c % 3) * 0.4f + 0.1f);
        mat.roughness = 0.3f + 0.6f * h;
        mesh.materials.push_back(mat);
    }
    struct Tri { glm::vec3 p[3]; i32 material; };
    std::vector<Tri> all;
    const float step = edge / static_cast<float>(n);
    for (int c = 0; c < copies; ++c)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const glm::vec3 a = pos + glm::vec3(step * static_cast<float>(i), 0.0f, step * static_cast<float>(j));
                const glm::vec3 b = a + glm::vec3(step, 0.0f, 0.0f), d = a + glm::vec3(0.0f, 0.0f, step), e = a + glm::vec3(step, 0.0f, step);
                all.push_back({{a, d, b}, c});
                all.push_back({{b, d, e}, c});
            }
    for (size_t k = all.size(); k > 1; --k) std::swap(all[k - 1], all[next() % k]);   // Fisher-Yates
    for (const Tri &t : all) {
        const i32 base = static_cast<i32>(mesh.vertices.size());
        for (const glm::vec3 &p : t.p) {
            MeshData::Vertex v;
            v.position = p;
            v.normal = glm::vec3(0.0f, 1.0f, 0.0f);
            v.uv = glm::vec2(p.x, p.z);
            mesh.vertices.push_back(v);
        }
        MeshData::Identifier id;
        id.index = glm::ivec3(base, base + 1, base + 2);
        id.materialIndex = t.material;
        mesh.identifiers.push_back(id);
    }
    mesh.deriveTextureFlags();
    return mesh;
}

}  // namespace procedural
