#include "obj_loader.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <unordered_map>

namespace orion_amd {

namespace {

struct Corner { int v, vt, vn; };  // 0-based, -1 when absent

std::string dirname_of(const std::string& p) {
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
}

std::string rest_of_line(std::istringstream& in) {
    std::string s;
    std::getline(in, s);
    size_t a = s.find_first_not_of(" \t\r");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r");
    return s.substr(a, b - a + 1);
}

void load_mtl(const std::string& path, std::vector<Material>& mats, std::map<std::string, uint32_t>& by_name) {
    std::ifstream f(path);
    if (!f) {
        std::printf("Error opening material file %s\n", path.c_str());
        return;
    }
    std::string line;
    Material* cur = nullptr;
    while (std::getline(f, line)) {
        size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        std::istringstream in(line);
        std::string key;
        if (!(in >> key)) continue;
        if (key == "newmtl") {
            Material m;
            m.name = rest_of_line(in);
            by_name[m.name] = (uint32_t)mats.size();
            mats.push_back(m);
            cur = &mats.back();
        } else if (!cur) {
            continue;
        } else if (key == "Ka") { in >> cur->ambient[0] >> cur->ambient[1] >> cur->ambient[2];
        } else if (key == "Kd") { in >> cur->diffuse[0] >> cur->diffuse[1] >> cur->diffuse[2];
        } else if (key == "Ks") { in >> cur->specular[0] >> cur->specular[1] >> cur->specular[2];
        } else if (key == "Ke") { in >> cur->emissive[0] >> cur->emissive[1] >> cur->emissive[2];
        } else if (key == "Ns") { in >> cur->shininess;
        } else if (key == "d") { in >> cur->opacity;
        } else if (key == "map_Kd") { cur->map_diffuse = rest_of_line(in);
        } else if (key == "map_Ks") { cur->map_specular = rest_of_line(in);
        } else if (key == "map_Ka") { cur->map_ambient = rest_of_line(in);
        } else if (key == "map_Bump" || key == "map_bump" || key == "bump") { cur->map_bump = rest_of_line(in);
        }
    }
}

// "7", "7/2", "7//3", "7/2/3", negative = relative to the current count.
Corner parse_corner(const std::string& tok, size_t nv, size_t nvt, size_t nvn) {
    Corner c = {-1, -1, -1};
    int vals[3] = {0, 0, 0};
    bool has[3] = {false, false, false};
    size_t pos = 0;
    for (int k = 0; k < 3 && pos <= tok.size(); k++) {
        size_t slash = tok.find('/', pos);
        std::string part = tok.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
        if (!part.empty()) { vals[k] = std::atoi(part.c_str()); has[k] = true; }
        if (slash == std::string::npos) break;
        pos = slash + 1;
    }
    auto resolve = [](int idx, size_t count) { return idx > 0 ? idx - 1 : (int)count + idx; };
    if (has[0]) c.v = resolve(vals[0], nv);
    if (has[1]) c.vt = resolve(vals[1], nvt);
    if (has[2]) c.vn = resolve(vals[2], nvn);
    return c;
}

struct VertexKey {
    uint32_t w[8];
    bool operator==(const VertexKey& o) const { return std::memcmp(w, o.w, sizeof w) == 0; }
};
struct VertexKeyHash {
    size_t operator()(const VertexKey& k) const {
        uint64_t h = 1469598103934665603ull;
        for (uint32_t x : k.w) { h ^= x; h *= 1099511628211ull; }
        return (size_t)h;
    }
};

struct MeshBuilder {
    Mesh mesh;
    std::unordered_map<VertexKey, uint32_t, VertexKeyHash> seen;
    void add(const Vertex& v) {
        VertexKey k;
        std::memcpy(k.w, &v, sizeof k.w);
        auto it = seen.find(k);
        if (it == seen.end()) {
            it = seen.emplace(k, (uint32_t)mesh.vertices.size()).first;
            mesh.vertices.push_back(v);
        }
        mesh.indices.push_back(it->second);
    }
};

} // namespace

bool load_obj(const std::string& path, Model& out) {
    out = Model();
    std::ifstream f(path);
    if (!f) {
        std::printf("ERROR::OBJ:: cannot open %s\n", path.c_str());
        return false;
    }
    out.directory = dirname_of(path);

    std::vector<float> pos, tex, nrm;  // 3, 2, 3 floats per entry
    std::map<std::string, uint32_t> mat_by_name;
    std::vector<MeshBuilder> builders;
    MeshBuilder cur;
    std::string group_name;
    bool have_material = false;

    auto flush = [&]() {
        const uint32_t keep = cur.mesh.material;
        if (!cur.mesh.indices.empty()) builders.push_back(std::move(cur));
        cur = MeshBuilder();
        cur.mesh.material = keep;
        cur.mesh.name = group_name;
    };

    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream in(line);
        std::string key;
        if (!(in >> key)) continue;
        if (key == "v") {
            float x = 0, y = 0, z = 0; in >> x >> y >> z;
            pos.push_back(x); pos.push_back(y); pos.push_back(z);
        } else if (key == "vt") {
            float u = 0, v = 0; in >> u >> v;
            tex.push_back(u); tex.push_back(v);
        } else if (key == "vn") {
            float x = 0, y = 0, z = 0; in >> x >> y >> z;
            nrm.push_back(x); nrm.push_back(y); nrm.push_back(z);
        } else if (key == "mtllib") {
            load_mtl(out.directory + "/" + rest_of_line(in), out.materials, mat_by_name);
        } else if (key == "g" || key == "o") {
            group_name = rest_of_line(in);
            flush();
        } else if (key == "usemtl") {
            std::string name = rest_of_line(in);
            auto it = mat_by_name.find(name);
            uint32_t id;
            if (it == mat_by_name.end()) {
                std::printf("Warning: material '%s' not found, using defaults\n", name.c_str());
                Material m; m.name = name;
                id = (uint32_t)out.materials.size();
                mat_by_name[name] = id;
                out.materials.push_back(m);
            } else {
                id = it->second;
            }
            if (!have_material || id != cur.mesh.material) flush();
            cur.mesh.material = id;
            have_material = true;
            if (cur.mesh.name.empty()) cur.mesh.name = name;
        } else if (key == "f") {
            std::vector<Corner> corners;
            std::string tok;
            while (in >> tok) corners.push_back(parse_corner(tok, pos.size() / 3, tex.size() / 2, nrm.size() / 3));
            if (corners.size() < 3) continue;
            for (size_t k = 1; k + 1 < corners.size(); k++) {
                const Corner tri[3] = {corners[0], corners[k], corners[k + 1]};
                Vertex vx[3];
                bool need_normal = false;
                for (int c = 0; c < 3; c++) {
                    std::memset(&vx[c], 0, sizeof(Vertex));
                    if (tri[c].v < 0 || (size_t)tri[c].v >= pos.size() / 3) {
                        std::printf("ERROR::OBJ:: vertex index out of range in %s\n", path.c_str());
                        out = Model();
                        return false;
                    }
                    std::memcpy(vx[c].position, &pos[3 * tri[c].v], 12);
                    if (tri[c].vt >= 0 && (size_t)tri[c].vt < tex.size() / 2) {
                        vx[c].uv[0] = tex[2 * tri[c].vt];
                        vx[c].uv[1] = 1.0f - tex[2 * tri[c].vt + 1];
                    }
                    if (tri[c].vn >= 0 && (size_t)tri[c].vn < nrm.size() / 3) std::memcpy(vx[c].normal, &nrm[3 * tri[c].vn], 12);
                    else need_normal = true;
                }
                if (need_normal) {
                    const float* a = vx[0].position; const float* b = vx[1].position; const float* c = vx[2].position;
                    float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
                    float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                    float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    if (len > 0.0f) { n[0] /= len; n[1] /= len; n[2] /= len; }
                    for (int cidx = 0; cidx < 3; cidx++) std::memcpy(vx[cidx].normal, n, 12);
                }
                for (int c = 0; c < 3; c++) cur.add(vx[c]);
            }
        }
    }
    flush();
    if (out.materials.empty()) out.materials.push_back(Material());
    for (MeshBuilder& b : builders) out.meshes.push_back(std::move(b.mesh));
    return true;
}

} // namespace orion_amd
