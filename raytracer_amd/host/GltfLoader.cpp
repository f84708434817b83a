// Binary glTF (.glb) ingestion: MeshData::LoadMeshData.
//
// The reference delegates parsing to tinygltf and image decoding to stb_image
// (reference src/MeshData.cpp:419-590); neither is available here, so this file
// carries a small JSON reader and the GLB container parse (the image decoders are PngDecode.cpp
// and JpegDecode.cpp) and then builds the MeshData the way the reference does:
//   * node hierarchy walked depth first, world = parent * local, local from
//     `matrix` or translate * rotate * scale (src/MeshData.cpp:11-45,169-197);
//   * positions = world * (p,1); normals = normalize(transpose(inverse(mat3(world))) * n),
//     default normal (0,1,0); TEXCOORD_0 as float / u8 normalised / u16 normalised
//     (src/MeshData.cpp:47-124); positions and normals are read as tightly packed
//     floats, UVs honour the buffer view's stride, exactly like the reference;
//   * indexed primitives only produce triangles; indices u8/u16/u32 (src/MeshData.cpp:126-166);
//   * primitives whose material has KHR_materials_emissive_strength > 0 and a non-zero
//     emissiveFactor go to the light set, which is moved to the front (src/MeshData.cpp:180-188,444-454);
//   * materials from pbrMetallicRoughness + the KHR volume/emissive_strength/transmission/ior
//     extensions (src/MeshData.cpp:199-323);
//   * textures: 8-bit images become k/255.0f texels (src/MeshData.cpp:475-528).  Images are
//     decoded to RGBA like tinygltf's default loader does; PNG and Huffman-coded 8-bit JPEG are
//     read — other encodings leave the texture without data, with a message.
#include "MeshData.h"
#include "ImageIO.h"

#include "glm/gtc/matrix_transform.hpp"
#include "glm/gtc/quaternion.hpp"
#include "glm/gtc/type_ptr.hpp"


#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <vector>

namespace {

// ---- minimal JSON -------------------------------------------------------------------
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json *find(const std::string &key) const {
        if (kind != Object) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const std::string &key) const { return find(key) != nullptr; }
    const Json &at(const std::string &key) const {
        static const Json none;
        const Json *j = find(key);
        return j ? *j : none;
    }
    const Json &at(size_t i) const {
        static const Json none;
        return (kind == Array && i < arr.size()) ? arr[i] : none;
    }
    size_t size() const { return kind == Array ? arr.size() : 0; }
    bool isNumber() const { return kind == Number; }
    double number(double fallback = 0) const { return kind == Number ? num : fallback; }
    int integer(int fallback = -1) const { return kind == Number ? static_cast<int>(num) : fallback; }
    std::string text(const std::string &fallback = "") const { return kind == String ? str : fallback; }
};

class JsonReader {
    const char *p, *end;
    bool ok = true;

    void skip() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
    }
    bool eat(char c) {
        skip();
        if (p < end && *p == c) { ++p; return true; }
        return false;
    }
    std::string parseString() {
        std::string out;
        if (!eat('"')) { ok = false; return out; }
        while (p < end && *p != '"') {
            char c = *p++;
            if (c == '\\' && p < end) {
                char e = *p++;
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        unsigned code = 0;
                        for (int i = 0; i < 4 && p < end; ++i) code = code * 16 + static_cast<unsigned>(std::strtol(std::string(1, *p++).c_str(), nullptr, 16));
                        if (code < 0x80) out += static_cast<char>(code);
                        else if (code < 0x800) { out += static_cast<char>(0xC0 | (code >> 6)); out += static_cast<char>(0x80 | (code & 0x3F)); }
                        else { out += static_cast<char>(0xE0 | (code >> 12)); out += static_cast<char>(0x80 | ((code >> 6) & 0x3F)); out += static_cast<char>(0x80 | (code & 0x3F)); }
                        break;
                    }
                    default: out += e; break;
                }
            } else {
                out += c;
            }
        }
        if (p < end) ++p; else ok = false;
        return out;
    }
    Json parseValue(int depth) {
        Json v;
        skip();
        if (p >= end || depth > 64) { ok = false; return v; }
        if (*p == '{') {
            ++p;
            v.kind = Json::Object;
            if (eat('}')) return v;
            do {
                std::string key = parseString();
                if (!eat(':')) { ok = false; return v; }
                v.obj.emplace_back(std::move(key), parseValue(depth + 1));
            } while (ok && eat(','));
            if (!eat('}')) ok = false;
        } else if (*p == '[') {
            ++p;
            v.kind = Json::Array;
            if (eat(']')) return v;
            do { v.arr.push_back(parseValue(depth + 1)); } while (ok && eat(','));
            if (!eat(']')) ok = false;
        } else if (*p == '"') {
            v.kind = Json::String;
            v.str = parseString();
        } else if (!std::strncmp(p, "true", std::min<size_t>(4, end - p))) {
            v.kind = Json::Bool; v.b = true; p += 4;
        } else if (!std::strncmp(p, "false", std::min<size_t>(5, end - p))) {
            v.kind = Json::Bool; v.b = false; p += 5;
        } else if (!std::strncmp(p, "null", std::min<size_t>(4, end - p))) {
            p += 4;
        } else {
            char *stop = nullptr;
            std::string tmp(p, std::min<size_t>(64, end - p));
            v.num = std::strtod(tmp.c_str(), &stop);
            if (stop == tmp.c_str()) { ok = false; return v; }
            v.kind = Json::Number;
            p += stop - tmp.c_str();
        }
        return v;
    }

public:
    JsonReader(const char *b, const char *e) : p(b), end(e) {}
    bool parse(Json &out) {
        out = parseValue(0);
        return ok;
    }
};

// ---- glTF document helpers ------------------------------------------------------------
struct Glb {
    Json doc;
    std::vector<unsigned char> bin;
};

bool readGlb(const std::string &path, Glb &glb, std::string &err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open file"; return false; }
    std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto le32 = [&](size_t at) { return uint32_t(bytes[at]) | (uint32_t(bytes[at + 1]) << 8) | (uint32_t(bytes[at + 2]) << 16) | (uint32_t(bytes[at + 3]) << 24); };
    if (bytes.size() < 20 || le32(0) != 0x46546C67u) { err = "not a binary glTF file"; return false; }
    if (le32(4) != 2u) { err = "unsupported glTF version"; return false; }
    size_t at = 12;
    bool haveJson = false;
    while (at + 8 <= bytes.size()) {
        const uint32_t len = le32(at), type = le32(at + 4);
        if (at + 8 + len > bytes.size()) { err = "truncated chunk"; return false; }
        if (type == 0x4E4F534Au) {   // JSON
            JsonReader reader(reinterpret_cast<const char *>(&bytes[at + 8]), reinterpret_cast<const char *>(&bytes[at + 8]) + len);
            if (!reader.parse(glb.doc)) { err = "malformed JSON chunk"; return false; }
            haveJson = true;
        } else if (type == 0x004E4942u && glb.bin.empty()) {   // BIN
            glb.bin.assign(bytes.begin() + at + 8, bytes.begin() + at + 8 + len);
        }
        at += 8 + ((len + 3u) & ~3u);
    }
    if (!haveJson) { err = "no JSON chunk"; return false; }
    return true;
}

struct AccessorView {
    const unsigned char *data = nullptr;  // first element
    size_t count = 0;
    size_t stride = 0;       // accessor.ByteStride(view)
    int componentType = 0;
    int width = 0;           // components per element (SCALAR 1 ... MAT4 16)
    bool packedFits = false; // `count` tightly packed elements from `data` also lie inside the buffer
    bool ok = false;
};

int componentSize(int componentType) {
    switch (componentType) {
        case 5120: case 5121: return 1;
        case 5122: case 5123: return 2;
        case 5125: case 5126: return 4;
        default: return 0;
    }
}
int typeCount(const std::string &t) {
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4") return 4;
    if (t == "MAT4") return 16;
    return 0;
}

// A byte quantity from the file: a finite number in [0, limit], else `bad`.
size_t byteField(const Json &j, double fallback, size_t limit, bool &bad) {
    const double v = j.number(fallback);
    if (!(v >= 0.0) || v > static_cast<double>(limit)) {
        bad = true;
        return 0;
    }
    return static_cast<size_t>(v);
}

// The file is the trust boundary: every offset, count and stride is range-checked against the
// binary chunk before anything is read through the view.
AccessorView viewOf(const Glb &glb, int accessorIndex) {
    AccessorView v;
    if (accessorIndex < 0) return v;
    const Json &acc = glb.doc.at("accessors").at(static_cast<size_t>(accessorIndex));
    if (acc.kind != Json::Object) return v;
    const int viewIndex = acc.at("bufferView").integer(-1);
    if (viewIndex < 0) return v;
    const Json &bv = glb.doc.at("bufferViews").at(static_cast<size_t>(viewIndex));
    if (bv.kind != Json::Object || bv.at("buffer").integer(0) != 0) return v;
    const size_t size = glb.bin.size();
    bool bad = false;
    const size_t viewOffset = byteField(bv.at("byteOffset"), 0, size, bad), accOffset = byteField(acc.at("byteOffset"), 0, size, bad);
    const size_t viewStride = byteField(bv.at("byteStride"), 0, size, bad);
    v.count = byteField(acc.at("count"), 0, size, bad);   // an element takes at least one byte
    if (bad || viewOffset + accOffset > size) return v;
    const size_t offset = viewOffset + accOffset;
    v.componentType = acc.at("componentType").integer(0);
    v.width = typeCount(acc.at("type").text());
    const size_t elem = static_cast<size_t>(componentSize(v.componentType)) * static_cast<size_t>(v.width);
    v.stride = viewStride ? viewStride : elem;
    if (elem == 0 || v.count == 0 || elem > size - offset) return v;
    if (v.count - 1 > (size - offset - elem) / v.stride) return v;
    v.packedFits = v.count <= (size - offset) / elem;
    v.data = glb.bin.data() + offset;
    v.ok = true;
    return v;
}

glm::mat4 localTransform(const Json &node) {
    glm::mat4 transform(1.0f);
    const Json &m = node.at("matrix");
    if (m.size() == 16) {
        double raw[16];
        for (size_t i = 0; i < 16; ++i) raw[i] = m.at(i).number();
        return glm::make_mat4(raw);
    }
    const Json &t = node.at("translation"), &r = node.at("rotation"), &s = node.at("scale");
    if (t.size() == 3) transform = glm::translate(transform, glm::vec3(t.at(0).number(), t.at(1).number(), t.at(2).number()));
    if (r.size() == 4) {
        glm::quat q(static_cast<float>(r.at(3).number()), static_cast<float>(r.at(0).number()), static_cast<float>(r.at(1).number()),
                    static_cast<float>(r.at(2).number()));
        transform *= glm::mat4_cast(q);
    }
    if (s.size() == 3) transform = glm::scale(transform, glm::vec3(s.at(0).number(), s.at(1).number(), s.at(2).number()));
    return transform;
}

void appendPrimitive(MeshData &mesh, const Glb &glb, const Json &prim, const glm::mat4 &world) {
    const glm::mat3 normalMatrix = glm::transpose(glm::inverse(glm::mat3(world)));
    const Json &attrs = prim.at("attributes");

    const AccessorView pos = viewOf(glb, attrs.at("POSITION").integer(-1));
    // read below as tightly packed float3, as the reference reads it: anything else is not a position stream
    if (!pos.ok || pos.componentType != 5126 || pos.width != 3 || !pos.packedFits) return;
    const size_t vertexCount = pos.count;
    const size_t base = mesh.vertices.size();
    mesh.vertices.resize(base + vertexCount);
    const float *p = reinterpret_cast<const float *>(pos.data);   // tightly packed, as the reference reads it
    for (size_t i = 0; i < vertexCount; ++i)
        mesh.vertices[base + i].position = glm::vec3(world * glm::vec4(p[i * 3 + 0], p[i * 3 + 1], p[i * 3 + 2], 1.0f));

    const AccessorView nrm = attrs.has("NORMAL") ? viewOf(glb, attrs.at("NORMAL").integer(-1)) : AccessorView();
    if (nrm.ok && nrm.count >= vertexCount && nrm.componentType == 5126 && nrm.width == 3 && nrm.packedFits) {
        const float *n = reinterpret_cast<const float *>(nrm.data);
        for (size_t i = 0; i < vertexCount; ++i)
            mesh.vertices[base + i].normal = glm::normalize(normalMatrix * glm::vec3(n[i * 3 + 0], n[i * 3 + 1], n[i * 3 + 2]));
    } else {
        for (size_t i = 0; i < vertexCount; ++i) mesh.vertices[base + i].normal = glm::normalize(normalMatrix * glm::vec3(0, 1, 0));
    }

    const AccessorView uv = attrs.has("TEXCOORD_0") ? viewOf(glb, attrs.at("TEXCOORD_0").integer(-1)) : AccessorView();
    if (uv.ok && uv.count >= vertexCount && uv.width >= 2) {
        for (size_t i = 0; i < vertexCount; ++i) {
            const unsigned char *e = uv.data + i * uv.stride;
            glm::vec2 &dst = mesh.vertices[base + i].uv;
            if (uv.componentType == 5126) {
                float f[2];
                std::memcpy(f, e, sizeof f);
                dst = glm::vec2(f[0], f[1]);
            } else if (uv.componentType == 5121) {
                dst = glm::vec2(e[0] / 255.0f, e[1] / 255.0f);
            } else if (uv.componentType == 5123) {
                u16 h[2];
                std::memcpy(h, e, sizeof h);
                dst = glm::vec2(h[0] / 65535.0f, h[1] / 65535.0f);
            }
        }
    }

    const int indicesAccessor = prim.at("indices").integer(-1);
    if (indicesAccessor < 0) return;   // non-indexed primitives contribute vertices but no triangles
    const AccessorView idx = viewOf(glb, indicesAccessor);
    if (!idx.ok || idx.width != 1 || !idx.packedFits || idx.componentType == 5126 || idx.componentType == 5120 || idx.componentType == 5122) return;
    const int material = prim.at("material").integer(-1);
    const int materialIndex = material < 0 ? 0 : material;
    const size_t triangles = idx.count / 3;
    mesh.identifiers.reserve(mesh.identifiers.size() + triangles);
    const int esize = componentSize(idx.componentType);
    for (size_t t = 0; t < triangles; ++t) {
        int v[3];
        for (int k = 0; k < 3; ++k) {
            const unsigned char *e = idx.data + (t * 3 + k) * static_cast<size_t>(esize);   // tightly packed
            if (esize == 1) v[k] = e[0];
            else if (esize == 2) { u16 h; std::memcpy(&h, e, 2); v[k] = h; }
            else { u32 w; std::memcpy(&w, e, 4); v[k] = w < 0x7fffffffu ? static_cast<int>(w) : -1; }
        }
        // a triangle that names a vertex the primitive does not have is dropped
        if (v[0] < 0 || v[1] < 0 || v[2] < 0 || static_cast<size_t>(v[0]) >= vertexCount || static_cast<size_t>(v[1]) >= vertexCount ||
            static_cast<size_t>(v[2]) >= vertexCount)
            continue;
        MeshData::Identifier iden;
        iden.index = glm::ivec3(v[0], v[1], v[2]) + glm::ivec3(static_cast<int>(base));
        iden.materialIndex = materialIndex;
        mesh.identifiers.push_back(iden);
    }
}

bool isLightMaterial(const Json &material) {
    const Json &ext = material.at("extensions").at("KHR_materials_emissive_strength");
    if (ext.kind != Json::Object) return false;
    const Json &strength = ext.at("emissiveStrength");
    const Json &e = material.at("emissiveFactor");
    // tinygltf fills a missing emissiveFactor with zeros.
    const double sum = e.size() == 3 ? e.at(0).number() + e.at(1).number() + e.at(2).number() : 0.0;
    return strength.isNumber() && strength.number() > 0 && sum > 0;
}

void walkNode(const Glb &glb, int nodeIndex, const glm::mat4 &parent, MeshData &mesh, MeshData &lights, int depth) {
    const Json &node = glb.doc.at("nodes").at(static_cast<size_t>(nodeIndex));
    if (node.kind != Json::Object || depth > 256) return;
    const glm::mat4 world = parent * localTransform(node);
    const int meshIndex = node.at("mesh").integer(-1);
    if (meshIndex >= 0) {
        const Json &prims = glb.doc.at("meshes").at(static_cast<size_t>(meshIndex)).at("primitives");
        for (size_t i = 0; i < prims.size(); ++i) {
            const Json &prim = prims.at(i);
            ASSERT(prim.at("mode").integer(4) == 4);   // triangles only
            const int material = prim.at("material").integer(-1);
            const Json &m = glb.doc.at("materials").at(static_cast<size_t>(material < 0 ? 0 : material));
            if (material >= 0 && isLightMaterial(m)) appendPrimitive(lights, glb, prim, world);
            else appendPrimitive(mesh, glb, prim, world);
        }
    }
    const Json &children = node.at("children");
    for (size_t i = 0; i < children.size(); ++i) walkNode(glb, children.at(i).integer(-1), world, mesh, lights, depth + 1);
}

Material convertMaterial(const Json &m) {
    Material out;
    const Json &pbr = m.at("pbrMetallicRoughness");
    out.texture.baseColorTexture = pbr.at("baseColorTexture").at("index").integer(-1);
    out.texture.metallicRoughnessTexture = pbr.at("metallicRoughnessTexture").at("index").integer(-1);
    out.texture.normalTexture = m.at("normalTexture").at("index").integer(-1);
    out.normalScale = static_cast<float>(m.at("normalTexture").at("scale").number(1.0));
    out.texture.emissiveTexture = m.at("emissiveTexture").at("index").integer(-1);
    out.texture.occlusionTexture = m.at("occlusionTexture").at("index").integer(-1);
    out.occlusionStrength = static_cast<float>(m.at("occlusionTexture").at("strength").number(1.0));

    const std::string alphaMode = m.at("alphaMode").text("OPAQUE");
    out.alphaCut = (alphaMode == "CUTOFF" || alphaMode == "MASK") ? static_cast<float>(m.at("alphaCutoff").number(0.5)) : 0.0f;

    const Json &base = pbr.at("baseColorFactor");
    if (base.size() >= 3) out.albedo = glm::vec3(base.at(0).number(), base.at(1).number(), base.at(2).number());

    const Json &volume = m.at("extensions").at("KHR_materials_volume").at("thicknessFactor");
    if (volume.isNumber()) out.subsurface = static_cast<float>(volume.number());

    out.metallic = static_cast<float>(pbr.at("metallicFactor").number(1.0));
    out.roughness = static_cast<float>(pbr.at("roughnessFactor").number(1.0));

    out.emissionColor = glm::vec3(0.0f);
    const Json &e = m.at("emissiveFactor");
    if (e.size() == 3) out.emissionColor = glm::vec3(e.at(0).number(), e.at(1).number(), e.at(2).number());
    out.emissionStrength = 1;
    const Json &strength = m.at("extensions").at("KHR_materials_emissive_strength").at("emissiveStrength");
    if (strength.isNumber()) out.emissionStrength = static_cast<float>(strength.number());

    const Json &transmission = m.at("extensions").at("KHR_materials_transmission");
    if (transmission.kind == Json::Object) {
        if (transmission.at("transmissionFactor").isNumber()) out.transmission = static_cast<float>(transmission.at("transmissionFactor").number());
        if (transmission.at("transmissionTexture").has("index")) out.texture.transmissionTexture = transmission.at("transmissionTexture").at("index").integer(-1);
    }
    const Json &ior = m.at("extensions").at("KHR_materials_ior").at("ior");
    if (ior.isNumber()) out.ior = static_cast<float>(ior.number());
    return out;
}

void convertTexture(const Glb &glb, const Json &texture, MeshData::Texture &out) {
    const Json &image = glb.doc.at("images").at(static_cast<size_t>(texture.at("source").integer(-1)));
    const Json &sampler = glb.doc.at("samplers").at(static_cast<size_t>(texture.at("sampler").integer(-1)));
    out.wrapS = sampler.at("wrapS").integer(10497);
    out.wrapT = sampler.at("wrapT").integer(10497);
    out.channelSize = 1;
    const int viewIndex = image.at("bufferView").integer(-1);
    if (viewIndex < 0) {
        std::cout << "\tImage without bufferView (external URI) is not loaded" << std::endl;
        return;
    }
    const Json &bv = glb.doc.at("bufferViews").at(static_cast<size_t>(viewIndex));
    bool bad = false;
    const size_t offset = byteField(bv.at("byteOffset"), 0, glb.bin.size(), bad), length = byteField(bv.at("byteLength"), 0, glb.bin.size(), bad);
    if (bad || bv.kind != Json::Object || length > glb.bin.size() - offset) return;
    int w = 0, h = 0;
    std::vector<unsigned char> rgba;
    const unsigned char *bytes = glb.bin.data() + offset;
    if (length >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
        std::string why;
        if (!imageio::DecodeJpegRGBA(bytes, length, w, h, rgba, why)) {
            std::cout << "\tJPEG image not decoded (" << why << "); texture left empty" << std::endl;
            return;
        }
    } else {
        std::string why;
        if (!imageio::DecodePngRGBA(bytes, length, w, h, rgba, why)) {
            std::cout << "\tImage not decoded (" << why << "; PNG and JPEG are the encodings read); texture left empty" << std::endl;
            return;
        }
    }
    if (w <= 0 || h <= 0 || rgba.size() != static_cast<size_t>(w) * static_cast<size_t>(h) * 4) return;
    out.width = w;
    out.height = h;
    out.channels = 4;
    out.data.resize(rgba.size());
    for (size_t i = 0; i < rgba.size(); ++i) out.data[i] = rgba[i] / 255.0f;
}

}  // namespace

MeshData MeshData::LoadMeshData(std::string modelPath) {
    Glb glb;
    std::string err;
    if (!readGlb(modelPath, glb, err)) {
        std::cerr << "Err: " << err << '\n' << "Failed to load " << modelPath << '\n';
        return {};
    }
    std::cout << "Load " << modelPath << " successfully!\n";

    MeshData mesh{}, lights{};
    const int sceneIndex = glb.doc.at("scene").integer(-1) > -1 ? glb.doc.at("scene").integer(0) : 0;
    const Json &roots = glb.doc.at("scenes").at(static_cast<size_t>(sceneIndex)).at("nodes");
    for (size_t i = 0; i < roots.size(); ++i) walkNode(glb, roots.at(i).integer(-1), glm::mat4(1.0f), mesh, lights, 0);

    mesh.adoptLights(lights);

    const Json &materials = glb.doc.at("materials");
    mesh.materials.reserve(materials.size());
    for (size_t i = 0; i < materials.size(); ++i) mesh.materials.push_back(convertMaterial(materials.at(i)));
    for (auto &id : mesh.identifiers)
        if (id.materialIndex < 0 || (!mesh.materials.empty() && id.materialIndex >= static_cast<i32>(mesh.materials.size()))) id.materialIndex = 0;
    for (auto &id : mesh.lightIdentifiers)
        if (id.materialIndex < 0 || (!mesh.materials.empty() && id.materialIndex >= static_cast<i32>(mesh.materials.size()))) id.materialIndex = 0;
    mesh.deriveTextureFlags();

    const Json &textures = glb.doc.at("textures");
    mesh.textures.resize(textures.size());
    for (size_t i = 0; i < textures.size(); ++i) {
        // An image the machine has no memory for leaves its texture empty, like one that does not decode.
        try {
            convertTexture(glb, textures.at(i), mesh.textures[i]);
        } catch (const std::bad_alloc &) {
            std::cout << "\tTexture " << i << ": out of memory while decoding; texture left empty" << std::endl;
            mesh.textures[i] = MeshData::Texture();
        }
    }
    return mesh;
}
