// Gltf.cpp — Scene::load_from_gltf for the target design.
//
// The reference's importer (Scene.cpp:82-214) is unfinished: the load check is
// inverted, only nodes[0] is visited, node TRS is ignored, normals are not
// transformed by the inverse transpose, load_gltf_materials has no body and no
// camera is derived.  This file completes those semantics with
// a small self-contained glTF 2.0 reader (GLB container or .gltf with a base64
// / external buffer); geometry flattening follows load_gltf_models
// (Scene.cpp:82-172): every primitive appended to sceneMergedMesh with
// world-space vertices, u16/u32 indices and the primitive's material index.
#include "Scene.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>

namespace rt {
namespace {

// ------------------------------------------------------------ tiny JSON DOM
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const char* key) const {
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json& operator[](const char* key) const {
        static const Json none;
        const Json* j = find(key);
        return j ? *j : none;
    }
    const Json& operator[](size_t i) const {
        static const Json none;
        return i < arr.size() ? arr[i] : none;
    }
    size_t size() const { return kind == Array ? arr.size() : 0; }
    bool has(const char* key) const { return find(key) != nullptr; }
    double number(double dflt) const { return kind == Number ? num : dflt; }
    int integer(int dflt) const { return kind == Number ? (int)num : dflt; }
};

class JsonParser {
public:
    JsonParser(const char* b, const char* e) : p(b), end(e) {}
    bool parse(Json* out) {
        bool ok = value(out);
        skip();
        return ok;
    }

private:
    const char *p, *end;
    void skip() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++;
    }
    bool literal(const char* s) {
        size_t n = std::strlen(s);
        if ((size_t)(end - p) < n || std::strncmp(p, s, n) != 0) return false;
        p += n;
        return true;
    }
    bool string(std::string* s) {
        if (p >= end || *p != '"') return false;
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                p++;
                switch (*p) {
                case 'n': s->push_back('\n'); break;
                case 't': s->push_back('\t'); break;
                case 'r': s->push_back('\r'); break;
                case 'b': s->push_back('\b'); break;
                case 'f': s->push_back('\f'); break;
                case 'u': // keep the low byte only; names are informational here
                    if (p + 4 < end) {
                        s->push_back((char)std::strtol(std::string(p + 1, p + 5).c_str(), nullptr, 16));
                        p += 4;
                    }
                    break;
                default: s->push_back(*p);
                }
                p++;
            } else {
                s->push_back(*p++);
            }
        }
        if (p >= end) return false;
        p++;
        return true;
    }
    bool value(Json* j) {
        skip();
        if (p >= end) return false;
        if (*p == '{') {
            j->kind = Json::Object;
            p++;
            skip();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                skip();
                std::string key;
                if (!string(&key)) return false;
                skip();
                if (p >= end || *p != ':') return false;
                p++;
                j->obj.emplace_back(key, Json());
                if (!value(&j->obj.back().second)) return false;
                skip();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; return true; }
                return false;
            }
        }
        if (*p == '[') {
            j->kind = Json::Array;
            p++;
            skip();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                j->arr.emplace_back();
                if (!value(&j->arr.back())) return false;
                skip();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return true; }
                return false;
            }
        }
        if (*p == '"') {
            j->kind = Json::String;
            return string(&j->str);
        }
        if (literal("true")) { j->kind = Json::Bool; j->b = true; return true; }
        if (literal("false")) { j->kind = Json::Bool; j->b = false; return true; }
        if (literal("null")) { j->kind = Json::Null; return true; }
        char* stop = nullptr;
        std::string tmp(p, std::min(end, p + 64));
        double v = std::strtod(tmp.c_str(), &stop);
        if (stop == tmp.c_str()) return false;
        p += stop - tmp.c_str();
        j->kind = Json::Number;
        j->num = v;
        return true;
    }
};

bool read_file(const std::string& path, std::vector<uint8_t>* out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0);
    out->resize((size_t)n);
    f.read((char*)out->data(), n);
    return (bool)f;
}

bool base64_decode(const std::string& s, size_t from, std::vector<uint8_t>* out) {
    uint32_t val = 0; // unsigned: the accumulator is shifted for the whole length of the text, only its low bits are read
    int bits = -8;
    for (size_t i = from; i < s.size(); i++) {
        unsigned char c = s[i];
        uint32_t d;
        if (c >= 'A' && c <= 'Z') d = c - 'A';
        else if (c >= 'a' && c <= 'z') d = c - 'a' + 26;
        else if (c >= '0' && c <= '9') d = c - '0' + 52;
        else if (c == '+') d = 62;
        else if (c == '/') d = 63;
        else if (c == '=') break;
        else continue;
        val = (val << 6) | d;
        bits += 6;
        if (bits >= 0) {
            out->push_back((uint8_t)((val >> bits) & 0xFF));
            bits -= 8;
        }
    }
    return true;
}

struct GltfDoc {
    Json root;
    std::vector<std::vector<uint8_t>> buffers;
};

struct AccessorView {
    const uint8_t* data = nullptr;
    size_t count = 0, stride = 0;
    int component = 0, width = 0; // width = components per element
};

bool view_accessor(const GltfDoc& doc, int index, AccessorView* out) {
    const Json& acc = doc.root["accessors"][(size_t)index];
    if (acc.kind != Json::Object) return false;
    const Json& bv = doc.root["bufferViews"][(size_t)acc["bufferView"].integer(-1)];
    if (bv.kind != Json::Object) return false;
    size_t buf = (size_t)bv["buffer"].integer(0);
    if (buf >= doc.buffers.size()) return false;
    const std::string& type = acc["type"].str;
    out->width = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
    out->component = acc["componentType"].integer(0);
    size_t csize = (out->component == 5126 || out->component == 5125) ? 4 : (out->component == 5123 || out->component == 5122) ? 2 : 1;
    out->count = (size_t)acc["count"].integer(0);
    out->stride = (size_t)bv["byteStride"].integer(0);
    if (out->stride == 0) out->stride = csize * out->width;
    size_t off = (size_t)acc["byteOffset"].integer(0) + (size_t)bv["byteOffset"].integer(0);
    if (out->width == 0 || off + (out->count ? (out->count - 1) * out->stride + csize * out->width : 0) > doc.buffers[buf].size()) return false;
    out->data = doc.buffers[buf].data() + off;
    return true;
}

mat4 node_local_matrix(const Json& node) {
    if (node["matrix"].size() == 16) { // Scene.cpp:178-181
        mat4 m;
        for (int i = 0; i < 16; i++) m.m[i] = (float)node["matrix"][(size_t)i].number(0);
        return m;
    }
    mat4 T(1.f), R(1.f), S(1.f);
    if (node["translation"].size() == 3) {
        T.m[12] = (float)node["translation"][(size_t)0].num;
        T.m[13] = (float)node["translation"][(size_t)1].num;
        T.m[14] = (float)node["translation"][(size_t)2].num;
    }
    if (node["rotation"].size() == 4)
        R = from_quat((float)node["rotation"][(size_t)0].num, (float)node["rotation"][(size_t)1].num, (float)node["rotation"][(size_t)2].num,
                      (float)node["rotation"][(size_t)3].num);
    if (node["scale"].size() == 3) {
        S.m[0] = (float)node["scale"][(size_t)0].num;
        S.m[5] = (float)node["scale"][(size_t)1].num;
        S.m[10] = (float)node["scale"][(size_t)2].num;
    }
    return T * R * S;
}

vec3 color_of(const Json& arr, vec3 dflt) {
    if (arr.size() < 3) return dflt;
    return vec3((float)arr[(size_t)0].num, (float)arr[(size_t)1].num, (float)arr[(size_t)2].num);
}

// What Scene::load_gltf_materials (Scene.h:108, no body at Scene.cpp:174) was to do.
Material* material_from_gltf(const Json& gm, size_t index, GltfMaterialMode mode, float emissive_scale) {
    vec3 emissive = color_of(gm["emissiveFactor"], vec3(0));
    if (emissive.x > 0 || emissive.y > 0 || emissive.z > 0) return new EmissiveMaterial(emissive_scale * emissive);
    const Json& pbr = gm["pbrMetallicRoughness"];
    vec3 base = color_of(pbr["baseColorFactor"], vec3(1));
    float metallic = (float)pbr["metallicFactor"].number(1.0); // glTF defaults
    float roughness = (float)pbr["roughnessFactor"].number(1.0);
    float transmission = (float)gm["extensions"]["KHR_materials_transmission"]["transmissionFactor"].number(0.0);
    switch (mode) {
    case GLTF_MATERIALS_MATTE: return new MatteMaterial(new ConstantTexture(base));
    case GLTF_MATERIALS_GGX:
        if (roughness > 0 && transmission == 0 && (index % 2 == 0)) return new Microfacet(new ConstantTexture(base), roughness, metallic);
        if (roughness > 0 && transmission == 0) return new BeckmannMicrofacet(new ConstantTexture(base), roughness, metallic);
        if (transmission > 0) return new RoughDielectric(1.5f, roughness, vec3(1), base);
        return new DisneyMetal(base, roughness, 0.f);
    default: {
        DisneyUber* u = new DisneyUber;
        u->base_color = base;
        u->metallic = metallic;
        u->roughness = roughness;
        u->specular_transmission = transmission;
        u->specular = 0.5f;
        u->ior = 1.5f;
        return u;
    }
    }
}

struct Importer {
    const GltfDoc& doc;
    Scene& scene;
    bool ok = true;

    void add_mesh(const Json& mesh, const mat4& toWorld) { // Scene::load_gltf_models, Scene.cpp:82-172
        Mesh& model = scene.sceneMergedMesh;
        const mat4 normalMatrix = transpose(inverse(toWorld)); // computed at Scene.cpp:84, now actually used
        const Json& prims = mesh["primitives"];
        for (size_t i = 0; i < prims.size(); i++) {
            const Json& prim = prims[i];
            if (prim["mode"].integer(4) != 4) continue; // triangles only
            AccessorView pos, nrm, uv, idx;
            if (!prim["attributes"].has("POSITION") || !view_accessor(doc, prim["attributes"]["POSITION"].integer(-1), &pos) ||
                pos.component != 5126 || pos.width != 3) {
                ok = false;
                continue;
            }
            bool has_n = prim["attributes"].has("NORMAL") && view_accessor(doc, prim["attributes"]["NORMAL"].integer(-1), &nrm) &&
                         nrm.component == 5126 && nrm.width == 3;
            bool has_uv = prim["attributes"].has("TEXCOORD_0") && view_accessor(doc, prim["attributes"]["TEXCOORD_0"].integer(-1), &uv) &&
                          uv.component == 5126 && uv.width == 2;
            // a malformed file must fail the load, not read past an accessor: attribute accessors hold one element per
            // vertex, indices come in threes and stay below the vertex count, the material index is not negative
            const int material = prim["material"].integer(0);
            if ((has_n && nrm.count < pos.count) || (has_uv && uv.count < pos.count) || material < 0) {
                ok = false;
                continue;
            }
            const uint32_t vertexStart = (uint32_t)model.verts.size();
            model.verts.resize(model.verts.size() + pos.count);
            for (size_t v = 0; v < pos.count; v++) {
                Vertex& vert = model.verts[vertexStart + v];
                float p[3];
                std::memcpy(p, pos.data + v * pos.stride, 12);
                vert.position = (toWorld * vec4(vec3(p[0], p[1], p[2]), 1.0f)).xyz();
                if (has_n) {
                    float n[3];
                    std::memcpy(n, nrm.data + v * nrm.stride, 12);
                    vert.normal = normalize((normalMatrix * vec4(vec3(n[0], n[1], n[2]), 0.0f)).xyz());
                }
                if (has_uv) {
                    float t[2];
                    std::memcpy(t, uv.data + v * uv.stride, 8);
                    vert.uv = vec2(t[0], t[1]);
                }
            }
            std::vector<uint32_t> indices;
            if (prim.has("indices")) {
                if (!view_accessor(doc, prim["indices"].integer(-1), &idx)) {
                    ok = false;
                    continue;
                }
                if (idx.component != 5125 && idx.component != 5123 && idx.component != 5121) {
                    ok = false;
                    model.verts.resize(vertexStart);
                    continue;
                }
                indices.resize(idx.count);
                for (size_t k = 0; k < idx.count; k++) {
                    const uint8_t* q = idx.data + k * idx.stride;
                    if (idx.component == 5125) { uint32_t t; std::memcpy(&t, q, 4); indices[k] = t; }
                    else if (idx.component == 5123) { uint16_t t; std::memcpy(&t, q, 2); indices[k] = t; }
                    else indices[k] = *q;
                }
            } else {
                indices.resize(pos.count);
                for (size_t k = 0; k < pos.count; k++) indices[k] = (uint32_t)k;
            }
            bool in_range = indices.size() % 3 == 0;
            for (size_t k = 0; in_range && k < indices.size(); k++) in_range = indices[k] < pos.count;
            if (!in_range) {
                ok = false;
                model.verts.resize(vertexStart);
                continue;
            }
            const size_t triStart = model.triangles.size();
            model.triangles.resize(triStart + indices.size() / 3);
            for (size_t t = 0; t < indices.size() / 3; t++) {
                MeshTriangle& tri = model.triangles[triStart + t];
                for (int c = 0; c < 3; c++) tri.indicies[c] = indices[3 * t + c] + vertexStart;
                tri.materialIndex = (uint32_t)material;
                if (!has_n) { // flat normal when the file has none
                    vec3 e1 = model.verts[tri.indicies[1]].position - model.verts[tri.indicies[0]].position;
                    vec3 e2 = model.verts[tri.indicies[2]].position - model.verts[tri.indicies[0]].position;
                    vec3 n = normalize(cross(e1, e2));
                    for (int c = 0; c < 3; c++) model.verts[tri.indicies[c]].normal = n;
                }
            }
        }
    }

    void visit(int nodeIndex, const mat4& parent, int depth) { // Scene::traverse_gltf_node_heirarchy, Scene.cpp:176-194
        if (depth > 64) return;
        const Json& node = doc.root["nodes"][(size_t)nodeIndex];
        if (node.kind != Json::Object) return;
        const mat4 transform = parent * node_local_matrix(node);
        if (node.has("mesh")) add_mesh(doc.root["meshes"][(size_t)node["mesh"].integer(-1)], transform);
        bool is_camera = node.has("camera");
        float yfov_deg = 39.6f; // frozen when the file carries no camera object
        if (is_camera) {
            const Json& cam = doc.root["cameras"][(size_t)node["camera"].integer(-1)];
            if (cam["perspective"].has("yfov")) yfov_deg = (float)(cam["perspective"]["yfov"].num * 180.0 / 3.14159265358979323846);
        }
        // Blender exports "Camera" -> "Camera_Orientation" even when the camera object itself is dropped
        if (!is_camera && node["name"].str == "Camera_Orientation" && !node.has("mesh")) is_camera = true;
        if (is_camera && !scene.hasCamera) {
            scene.sceneCamera = Camera(transform, yfov_deg, 512, 512);
            scene.sceneCameraView = transform;
            scene.sceneCameraFov = yfov_deg;
            scene.hasCamera = true;
        }
        const Json& kids = node["children"];
        for (size_t i = 0; i < kids.size(); i++) visit(kids[i].integer(-1), transform, depth + 1);
    }

};

} // namespace

bool Scene::load_from_gltf(const std::string& gltfPath) { // Scene.cpp:196-214
    std::vector<uint8_t> file;
    if (!read_file(gltfPath, &file)) {
        std::printf("Couldn't load gltf model: cannot read %s\n", gltfPath.c_str());
        return false;
    }
    GltfDoc doc;
    bool parsed = false;
    if (file.size() >= 20 && std::memcmp(file.data(), "glTF", 4) == 0) { // GLB container
        uint32_t jlen;
        std::memcpy(&jlen, file.data() + 12, 4);
        if (20 + (size_t)jlen > file.size()) {
            std::printf("Couldn't load gltf model: truncated GLB\n");
            return false;
        }
        JsonParser jp((const char*)file.data() + 20, (const char*)file.data() + 20 + jlen);
        parsed = jp.parse(&doc.root);
        size_t off = 20 + jlen;
        if (off + 8 <= file.size()) {
            uint32_t blen;
            std::memcpy(&blen, file.data() + off, 4);
            if (off + 8 + blen <= file.size()) doc.buffers.emplace_back(file.begin() + off + 8, file.begin() + off + 8 + blen);
        }
    } else {
        JsonParser jp((const char*)file.data(), (const char*)file.data() + file.size());
        parsed = jp.parse(&doc.root);
        const Json& bufs = doc.root["buffers"];
        std::string dir = gltfPath.substr(0, gltfPath.find_last_of("/\\") + 1);
        for (size_t i = 0; parsed && i < bufs.size(); i++) {
            const std::string& uri = bufs[i]["uri"].str;
            doc.buffers.emplace_back();
            size_t comma = uri.find(";base64,");
            if (uri.compare(0, 5, "data:") == 0 && comma != std::string::npos) base64_decode(uri, comma + 8, &doc.buffers.back());
            else if (!read_file(dir + uri, &doc.buffers.back())) parsed = false;
        }
    }
    if (!parsed || doc.root.kind != Json::Object) {
        std::printf("Couldn't load gltf model: malformed JSON\n");
        return false;
    }

    const Json& mats = doc.root["materials"];
    for (size_t i = 0; i < mats.size(); i++) materials.push_back(material_from_gltf(mats[i], i, gltf_material_mode, gltf_emissive_scale));
    if (materials.empty()) materials.push_back(new MatteMaterial(new ConstantTexture(0.8f)));

    Importer imp{doc, *this};
    const Json& scenes = doc.root["scenes"];
    const Json& roots = scenes[(size_t)doc.root["scene"].integer(0)]["nodes"]; // all roots, not nodes.at(0) (Scene.cpp:212)
    for (size_t i = 0; i < roots.size(); i++) imp.visit(roots[i].integer(-1), mat4(1.f), 0);
    if (!imp.ok) {
        std::printf("Couldn't load gltf model: unsupported or inconsistent accessor layout\n");
        return false;
    }
    for (MeshTriangle& t : sceneMergedMesh.triangles)
        if (t.materialIndex >= materials.size()) t.materialIndex = 0;
    return !sceneMergedMesh.triangles.empty();
}

} // namespace rt
