// Mini input-script driver: runs the command subset used by the reference's benchmark decks
// (/root/reference/example/simple/{sp,dp}.run) against the engine, so those files work unchanged on a
// machine that has no LAMMPS tree.  Inside a real LAMMPS build the same engine calls are made by the
// glue styles shown in INTEGRATION.md; this file stands in for LAMMPS' Input/ReadData/Velocity/Thermo.
//
//   dimension units boundary atom_style neighbor neigh_modify read_data mass run_style pair_style
//   pair_coeff compute velocity fix thermo_style thermo thermo_modify timestep run variable
//
// Semantics follow src/input.cpp (variable substitution ${name}, '#' comments, '&' continuation),
// src/read_data.cpp (header keywords, Masses/Atoms/Velocities sections), src/velocity.cpp:140-330
// (create ... loop all|local, dist uniform|gaussian, mom yes, rot no) with RanPark (src/random_park.cpp).
#include "engine.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <vector>

namespace meso {

namespace {

struct RanPark {
    int seed;
    int save = 0;
    double second = 0.0;
    explicit RanPark(int s) : seed(s) {}
    double uniform()
    {
        const int IA = 16807, IM = 2147483647, IQ = 127773, IR = 2836;
        int k = seed / IQ;
        seed = IA * (seed - k * IQ) - IR * k;
        if (seed < 0) seed += IM;
        return (1.0 / IM) * seed;
    }
    double gaussian()
    {
        double first;
        if (!save) {
            double v1, v2, rsq;
            do {
                v1 = 2.0 * uniform() - 1.0;
                v2 = 2.0 * uniform() - 1.0;
                rsq = v1 * v1 + v2 * v2;
            } while (rsq >= 1.0 || rsq == 0.0);
            double fac = std::sqrt(-2.0 * std::log(rsq) / rsq);
            second = v1 * fac;
            first = v2 * fac;
            save = 1;
        } else {
            first = second;
            save = 0;
        }
        return first;
    }
};

struct Deck {
    int natoms = 0, ntypes = 0;
    double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    int periodic[3] = {1, 1, 1};
    std::vector<double> x, v, mass;
    std::vector<int> tag, type;
    int nbonds = 0, nbondtypes = 0;
    bool bond_fene = false, pair_mini = false, pair_poly = false;
    int table_len = 0;
    std::vector<int> bond_i, bond_j, bond_t;
    int nangles = 0, nangletypes = 0;
    std::vector<int> ang_1, ang_2, ang_3, ang_t;
    bool bonds_sent = false;
    bool have_atoms = false, uploaded = false, is_setup = false;
    int thermo_every = 0;
    std::string atom_style, run_style;
    std::map<std::string, std::string> vars;
    std::map<std::string, std::string> computes;   // id -> style
    bool thermo_pe = false, thermo_press = false;
    double cpu_prev = 0.0;
    long step_prev = 0;
};

std::vector<std::string> split(const std::string &s)
{
    std::vector<std::string> w;
    std::istringstream is(s);
    std::string t;
    while (is >> t) w.push_back(t);
    return w;
}

int read_data(Engine &E, Deck &D, const std::string &path)
{
    std::ifstream f(path);
    if (!f) { E.err = "Cannot open file " + path; return 1; }
    std::string line;
    std::getline(f, line);   // title
    std::string section;
    bool have_v = false;
    while (std::getline(f, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line = line.substr(0, h);
        std::vector<std::string> w = split(line);
        if (w.empty()) continue;
        if (w.size() == 2 && w[1] == "atoms") { D.natoms = atoi(w[0].c_str()); continue; }
        if (w.size() == 3 && w[1] == "atom" && w[2] == "types") { D.ntypes = atoi(w[0].c_str()); continue; }
        if (w.size() == 4 && (w[2] == "xlo" || w[2] == "ylo" || w[2] == "zlo")) {
            int d = w[2][0] - 'x';
            D.lo[d] = atof(w[0].c_str());
            D.hi[d] = atof(w[1].c_str());
            continue;
        }
        if (w.size() == 2 && w[1] == "bonds") { D.nbonds = atoi(w[0].c_str()); continue; }
        if (w.size() == 3 && w[1] == "bond" && w[2] == "types") { D.nbondtypes = atoi(w[0].c_str()); continue; }
        if (w.size() == 2 && w[1] == "angles") { D.nangles = atoi(w[0].c_str()); continue; }
        if (w.size() == 3 && w[1] == "angle" && w[2] == "types") { D.nangletypes = atoi(w[0].c_str()); continue; }
        if (w.size() >= 2 && (w[1] == "dihedrals" || w[1] == "impropers")) continue;
        if (w.size() == 3 && w[2] == "types") continue;
        if (w[0] == "Bonds") {
            D.bond_i.resize(D.nbonds); D.bond_j.resize(D.nbonds); D.bond_t.resize(D.nbonds);
            for (int b = 0; b < D.nbonds; b++) {
                long id; int t, a1, a2;
                if (!(f >> id >> t >> a1 >> a2)) { E.err = "Unexpected end of data file"; return 1; }
                D.bond_t[b] = t; D.bond_i[b] = a1; D.bond_j[b] = a2;
            }
            continue;
        }
        if (w[0] == "Angles") {
            D.ang_1.resize(D.nangles); D.ang_2.resize(D.nangles); D.ang_3.resize(D.nangles); D.ang_t.resize(D.nangles);
            for (int b = 0; b < D.nangles; b++) {
                long id; int t, a1, a2, a3;
                if (!(f >> id >> t >> a1 >> a2 >> a3)) { E.err = "Unexpected end of data file"; return 1; }
                D.ang_t[b] = t; D.ang_1[b] = a1; D.ang_2[b] = a2; D.ang_3[b] = a3;
            }
            continue;
        }
        if (w[0] == "Masses" || w[0] == "Atoms" || w[0] == "Velocities") {
            section = w[0];
            if (section == "Atoms") {
                if (D.natoms <= 0) { E.err = "No atoms in data file header"; return 1; }
                D.x.assign((size_t)3 * D.natoms, 0.0);
                D.v.assign((size_t)3 * D.natoms, 0.0);
                D.tag.resize(D.natoms);
                D.type.resize(D.natoms);
                const bool molecular = D.atom_style == "dpd/bond/meso" || D.atom_style == "bond" || D.atom_style == "dpd/angle/meso" ||
                                       D.atom_style == "angle";
                for (int i = 0; i < D.natoms; i++) {
                    long id, mol = 0; int t; double a, b, c;
                    if (molecular) { if (!(f >> id >> mol >> t >> a >> b >> c)) { E.err = "Unexpected end of data file"; return 1; } }
                    else if (!(f >> id >> t >> a >> b >> c)) { E.err = "Unexpected end of data file"; return 1; }
                    std::getline(f, line);   // rest of line (image flags ignored)
                    if (t < 1 || t > D.ntypes) { E.err = "Invalid atom type in Atoms section of data file"; return 1; }
                    D.tag[i] = (int)id; D.type[i] = t;
                    D.x[3 * (size_t)i] = a; D.x[3 * (size_t)i + 1] = b; D.x[3 * (size_t)i + 2] = c;
                }
                D.have_atoms = true;
            } else if (section == "Masses") {
                D.mass.assign(D.ntypes + 1, 0.0);
                for (int t = 0; t < D.ntypes; t++) {
                    int id; double m;
                    if (!(f >> id >> m)) { E.err = "Unexpected end of data file"; return 1; }
                    if (id < 1 || id > D.ntypes) { E.err = "Invalid type for mass set"; return 1; }
                    D.mass[id] = m;
                }
            } else {
                if (!D.have_atoms) { E.err = "Must read Atoms before Velocities"; return 1; }
                std::map<int, int> idx;
                for (int i = 0; i < D.natoms; i++) idx[D.tag[i]] = i;
                for (int i = 0; i < D.natoms; i++) {
                    long id; double a, b, c;
                    if (!(f >> id >> a >> b >> c)) { E.err = "Unexpected end of data file"; return 1; }
                    auto it = idx.find((int)id);
                    if (it == idx.end()) { E.err = "Invalid atom ID in Velocities section of data file"; return 1; }
                    size_t m = it->second;
                    D.v[3 * m] = a; D.v[3 * m + 1] = b; D.v[3 * m + 2] = c;
                }
                have_v = true;
            }
            continue;
        }
    }
    (void)have_v;
    if (!D.have_atoms) { E.err = "No Atoms section in data file"; return 1; }
    return 0;
}

// Velocity::create (src/velocity.cpp:140-330): loop all, mom yes, rot no
int velocity_create(Engine &E, Deck &D, double t_desired, int seed, bool gaussian)
{
    if (seed <= 0) { E.err = "Illegal velocity create command"; return 1; }
    int n = D.natoms;
    std::vector<int> map(n + 1, -1);
    for (int i = 0; i < n; i++) {
        if (D.tag[i] < 1 || D.tag[i] > n) { E.err = "Atom IDs must be consecutive for velocity create loop all"; return 1; }
        map[D.tag[i]] = i;
    }
    RanPark rnd(seed);
    for (int i = 1; i <= n; i++) {
        double vx = gaussian ? rnd.gaussian() : rnd.uniform();
        double vy = gaussian ? rnd.gaussian() : rnd.uniform();
        double vz = gaussian ? rnd.gaussian() : rnd.uniform();
        size_t m = map[i];
        double factor = 1.0 / std::sqrt(D.mass[D.type[m]]);
        D.v[3 * m] = vx * factor; D.v[3 * m + 1] = vy * factor; D.v[3 * m + 2] = vz * factor;
    }
    double masstotal = 0.0, p[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) masstotal += D.mass[D.type[i]];
    for (int i = 0; i < n; i++) {
        double m = D.mass[D.type[i]];
        for (int d = 0; d < 3; d++) p[d] += D.v[3 * (size_t)i + d] * m;
    }
    if (masstotal > 0.0) for (int d = 0; d < 3; d++) p[d] /= masstotal;
    for (int i = 0; i < n; i++) for (int d = 0; d < 3; d++) D.v[3 * (size_t)i + d] -= p[d];
    double t = 0.0;
    for (int i = 0; i < n; i++) {
        const double *v = &D.v[3 * (size_t)i];
        t += (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) * D.mass[D.type[i]];
    }
    t /= (3.0 * n - 3.0);
    if (t == 0.0) { E.err = "Attempting to rescale a 0.0 temperature"; return 1; }
    double factor = std::sqrt(t_desired / t);
    for (size_t k = 0; k < D.v.size(); k++) D.v[k] *= factor;
    return 0;
}

int upload(Engine &E, Deck &D)
{
    if (D.uploaded) return 0;
    if (!D.have_atoms) { E.err = "Run command before simulation box is defined"; return 1; }
    int rc;
    if ((rc = E.set_box(D.lo, D.hi, D.periodic))) return rc;
    if (D.mass.empty()) { E.err = "All masses are not set"; return 1; }
    if ((rc = E.set_mass(D.ntypes, D.mass.data()))) return rc;
    if ((rc = E.atoms_upload(D.natoms, D.x.data(), D.v.data(), D.tag.data(), D.type.data(), nullptr, nullptr))) return rc;
    if (D.nbonds > 0 && (rc = E.bonds_upload(D.nbonds, D.bond_i.data(), D.bond_j.data(), D.bond_t.data()))) return rc;
    if (D.nangles > 0 && (rc = E.angles_upload(D.nangles, D.ang_1.data(), D.ang_2.data(), D.ang_3.data(), D.ang_t.data()))) return rc;
    D.uploaded = true;
    return 0;
}

double now()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

int script_run(Engine &E, const char *path, const char *var_name, const char *var_value, std::string &out)
{
    std::ifstream f(path);
    if (!f) { E.err = std::string("Cannot open input script ") + path; return 1; }
    Deck D;
    if (var_name && var_value) D.vars[var_name] = var_value;
    std::string dir(path);
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? "" : dir.substr(0, slash + 1);

    // pending pair_coeff lines are applied once the type count is known (after read_data)
    std::string line, cont;
    char buf[256];
    double t_run0 = 0.0;
    while (std::getline(f, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line = line.substr(0, h);
        while (!line.empty() && (line.back() == ' ' || line.back() == '\t' || line.back() == '\r')) line.pop_back();
        if (!line.empty() && line.back() == '&') { cont += line.substr(0, line.size() - 1) + " "; continue; }
        line = cont + line;
        cont.clear();
        // ${var} substitution (src/input.cpp Input::substitute)
        size_t p;
        while ((p = line.find("${")) != std::string::npos) {
            size_t q = line.find('}', p);
            if (q == std::string::npos) { E.err = "Invalid variable name"; return 1; }
            std::string name = line.substr(p + 2, q - p - 2);
            auto it = D.vars.find(name);
            if (it == D.vars.end()) { E.err = "Substitution for illegal variable " + name; return 1; }
            line = line.substr(0, p) + it->second + line.substr(q + 1);
        }
        std::vector<std::string> w = split(line);
        if (w.empty()) continue;
        const std::string &c = w[0];
        int rc = 0;
        if (c == "dimension") {
            if (w.size() != 2 || w[1] != "3") { E.err = "USER-MESO hot path supports dimension 3 only"; return 1; }
        } else if (c == "units") {
            if (w.size() != 2 || w[1] != "lj") { E.err = "Only units lj are supported by this driver"; return 1; }
        } else if (c == "boundary") {
            if (w.size() != 4) { E.err = "Illegal boundary command"; return 1; }
            for (int d = 0; d < 3; d++) {
                if (w[1 + d] == "p") D.periodic[d] = 1;
                else if (w[1 + d] == "f") D.periodic[d] = 0;
                else { E.err = "Illegal boundary command"; return 1; }
            }
        } else if (c == "atom_style") {
            if (w.size() < 2 || (w[1] != "dpd/atomic/meso" && w[1] != "atomic" && w[1] != "dpd/bond/meso" && w[1] != "bond" && w[1] != "dpd/angle/meso" && w[1] != "angle")) { E.err = "Invalid atom style " + (w.size() > 1 ? w[1] : ""); return 1; }
            D.atom_style = w[1];
        } else if (c == "variable") {
            if (w.size() >= 4 && (w[2] == "index" || w[2] == "equal" || w[2] == "string")) { if (!D.vars.count(w[1])) D.vars[w[1]] = w[3]; }
            else { E.err = "Illegal variable command"; return 1; }
        } else if (c == "neighbor") {
            if (w.size() != 3 || w[2] != "bin") { E.err = "Illegal neighbor command"; return 1; }
            D.vars["__skin"] = w[1];
        } else if (c == "neigh_modify") {
            for (size_t k = 1; k + 1 < w.size(); k += 2) {
                if (w[k] == "delay" || w[k] == "every" || w[k] == "check") D.vars["__" + w[k]] = w[k + 1];
                else { E.err = "Illegal neigh_modify command"; return 1; }
            }
        } else if (c == "read_data") {
            if (w.size() != 2) { E.err = "Illegal read_data command"; return 1; }
            std::string p2 = w[1];
            std::ifstream probe(p2);
            if (!probe) p2 = dir + w[1];
            if ((rc = read_data(E, D, p2))) return rc;
            snprintf(buf, sizeof buf, "  %d atoms\n", D.natoms);
            out += buf;
        } else if (c == "read_restart") {
            // the engine's own per-rank file (restart.hip): box, masses, pair / bond / angle coefficients, atoms and topology
            if (w.size() != 2) { E.err = "Illegal read_restart command"; return 1; }
            std::string p2 = w[1];
            { std::ifstream probe(p2); if (!probe) p2 = dir + w[1]; }
            if ((rc = E.read_restart(p2))) return rc;
            D.have_atoms = true; D.uploaded = true; D.is_setup = false;
            D.natoms = E.nlocal; D.ntypes = E.restart_ntypes(); D.mass = E.restart_masses();
            E.restart_box(D.lo, D.hi, D.periodic);
            snprintf(buf, sizeof buf, "  %d atoms\n", D.natoms);
            out += buf;
        } else if (c == "write_restart") {
            if (w.size() != 2) { E.err = "Illegal write_restart command"; return 1; }
            if ((rc = upload(E, D))) return rc;
            if ((rc = E.write_restart(w[1]))) return rc;
        } else if (c == "mass") {
            if (w.size() != 3) { E.err = "Illegal mass command"; return 1; }
            if (D.mass.empty()) D.mass.assign(D.ntypes + 1, 0.0);
            int t = atoi(w[1].c_str());
            if (t < 1 || t > D.ntypes) { E.err = "Invalid type for mass set"; return 1; }
            D.mass[t] = atof(w[2].c_str());
        } else if (c == "run_style") {
            if (w.size() != 2 || (w[1] != "mvv/meso" && w[1] != "verlet/meso")) { E.err = "Illegal run_style command"; return 1; }
            D.run_style = w[1];
        } else if (c == "pair_style") {
            if (w.size() != 4 || (w[1] != "dpd/meso" && w[1] != "dpd/fast/meso" && w[1] != "dpd/mini/meso" && w[1] != "dpd/polyforce/meso")) {
                // pair_style dpd/tableforce/meso cut_global seed table_length (pair_dpd_tableforce_meso.cu:284-291)
                if (w.size() == 5 && w[1] == "dpd/tableforce/meso") {
                    if (D.mass.empty()) { E.err = "pair_style before read_data"; return 1; }
                    if ((rc = E.set_mass(D.ntypes, D.mass.data()))) return rc;
                    D.pair_mini = D.pair_poly = false;
                    D.table_len = atoi(w[4].c_str());
                    if (D.table_len < 2) { E.err = "dpd/tableforce/meso command require: cut_global seed table_length"; return 1; }
                    if ((rc = E.pair_settings(4, atof(w[2].c_str()), atoi(w[3].c_str())))) return rc;
                    continue;
                }
                E.err = "Illegal pair_style command"; return 1;
            }
            D.table_len = 0;
            D.pair_mini = w[1] == "dpd/mini/meso";
            D.pair_poly = w[1] == "dpd/polyforce/meso";
            if (D.mass.empty()) { E.err = "pair_style before read_data"; return 1; }
            if ((rc = E.set_mass(D.ntypes, D.mass.data()))) return rc;
            if ((rc = E.pair_settings(D.pair_poly ? 3 : D.pair_mini ? 2 : (w[1] == "dpd/fast/meso" ? 1 : 0), atof(w[2].c_str()), atoi(w[3].c_str())))) return rc;
        } else if (c == "pair_coeff") {
            // dpd/mini/meso: pair_coeff * * a0 gamma sigma (pair_dpd_minimal_meso.cu:248-265); the others: ... s [rc]
            // dpd/polyforce/meso: pair_coeff i j gamma sigma order c_order ... c_0 (pair_dpd_polyforce_meso.cu:290-335)
            if (D.table_len ? (w.size() != 6 && (int)w.size() != 5 + D.table_len) : D.pair_poly ? (w.size() < 7 || (int)w.size() != 7 + atoi(w[5].c_str())) : D.pair_mini ? w.size() < 6 : (w.size() < 7 || w.size() > 8)) { E.err = "Incorrect args for pair coefficients"; return 1; }
            auto bounds = [&](const std::string &s, int &lo, int &hi) {
                if (s == "*") { lo = 1; hi = D.ntypes; return; }
                size_t star = s.find('*');
                if (star == std::string::npos) { lo = hi = atoi(s.c_str()); return; }
                lo = star == 0 ? 1 : atoi(s.substr(0, star).c_str());
                hi = star + 1 == s.size() ? D.ntypes : atoi(s.substr(star + 1).c_str());
            };
            int ilo, ihi, jlo, jhi;
            bounds(w[1], ilo, ihi);
            bounds(w[2], jlo, jhi);
            int count = 0;
            for (int i = ilo; i <= ihi; i++)
                for (int j = std::max(jlo, i); j <= jhi; j++) {
                    if (D.table_len) {
                        // pair_coeff i j gamma sigma < file | table_length values > (pair_dpd_tableforce_meso.cu:306-356)
                        std::vector<double> tb;
                        if (w.size() == 6) {
                            std::ifstream ft(w[5]);
                            if (!ft) { E.err = "Cannot open force table file for dpd/tableforce/meso"; return 1; }
                            double val;
                            while ((int)tb.size() < D.table_len && (ft >> val)) tb.push_back(val);
                            if ((int)tb.size() < D.table_len) { E.err = "Insufficient parameters in force table file for dpd/tableforce/meso"; return 1; }
                        } else {
                            for (size_t k = 5; k < w.size(); k++) tb.push_back(atof(w[k].c_str()));
                        }
                        if ((rc = E.pair_coeff_table(i, j, atof(w[3].c_str()), atof(w[4].c_str()), (int)tb.size(), tb.data()))) return rc;
                        count++;
                        continue;
                    }
                    if (D.pair_poly) {
                        std::vector<double> pc;
                        for (size_t k = 6; k < w.size(); k++) pc.push_back(atof(w[k].c_str()));
                        if ((rc = E.pair_coeff_poly(i, j, atof(w[3].c_str()), atof(w[4].c_str()), atoi(w[5].c_str()), pc.data()))) return rc;
                        count++;
                        continue;
                    }
                    if ((rc = E.pair_coeff(i, j, atof(w[3].c_str()), atof(w[4].c_str()), atof(w[5].c_str()),
                                           w.size() > 6 ? atof(w[6].c_str()) : 1.0, w.size() == 8 ? atof(w[7].c_str()) : 0.0))) return rc;
                    count++;
                }
            if (!count) { E.err = "Incorrect args for pair coefficients"; return 1; }
        } else if (c == "special_bonds") {
            if (w.size() != 5 || w[1] != "lj") { E.err = "Illegal special_bonds command"; return 1; }
            if ((rc = E.special_bonds(atof(w[2].c_str()), atof(w[3].c_str()), atof(w[4].c_str())))) return rc;
        } else if (c == "bond_style") {
            if (w.size() != 2 || (w[1] != "harmonic/meso" && w[1] != "fene/meso")) { E.err = "Invalid bond style"; return 1; }
            D.bond_fene = w[1] == "fene/meso";
            if ((rc = E.bond_style(std::max(D.nbondtypes, 1), D.bond_fene ? 1 : 0))) return rc;
        } else if (c == "bond_coeff") {
            if (w.size() != (D.bond_fene ? 6u : 4u)) { E.err = "Incorrect args for bond coefficients"; return 1; }
            if ((rc = E.bond_coeff(atoi(w[1].c_str()), atof(w[2].c_str()), atof(w[3].c_str()), D.bond_fene ? atof(w[4].c_str()) : 0.0,
                                   D.bond_fene ? atof(w[5].c_str()) : 0.0)))
                return rc;
        } else if (c == "angle_style") {
            if (w.size() != 2 || w[1] != "harmonic/meso") { E.err = "Invalid angle style"; return 1; }
            if ((rc = E.angle_style(std::max(D.nangletypes, 1)))) return rc;
        } else if (c == "angle_coeff") {
            if (w.size() != 4) { E.err = "Incorrect args for angle coefficients"; return 1; }
            if ((rc = E.angle_coeff(atoi(w[1].c_str()), atof(w[2].c_str()), atof(w[3].c_str())))) return rc;
        } else if (c == "compute") {
            if (w.size() < 4 || w[2] != "all") { E.err = "Illegal compute command"; return 1; }
            if (w[3] != "temp/meso" && w[3] != "pe/meso" && w[3] != "pressure/meso") { E.err = "Invalid compute style " + w[3]; return 1; }
            D.computes[w[1]] = w[3];
        } else if (c == "velocity") {
            if (w.size() < 5 || w[1] != "all" || w[2] != "create") { E.err = "Illegal velocity command"; return 1; }
            bool gauss = false;
            for (size_t k = 5; k + 1 < w.size(); k += 2) {
                if (w[k] == "loop") { if (w[k + 1] != "all") { E.err = "velocity loop " + w[k + 1] + " is not supported by this driver"; return 1; } }
                else if (w[k] == "dist") gauss = w[k + 1] == "gaussian";
                else if (w[k] == "mom" || w[k] == "rot") {}
                else { E.err = "Illegal velocity command"; return 1; }
            }
            if (D.mass.empty()) { E.err = "All masses are not set"; return 1; }
            if ((rc = velocity_create(E, D, atof(w[3].c_str()), atoi(w[4].c_str()), gauss))) return rc;
        } else if (c == "fix") {
            if (w.size() != 4 || w[2] != "all" || w[3] != "nve/meso") { E.err = "Invalid fix style"; return 1; }
        } else if (c == "thermo_style") {
            D.thermo_pe = D.thermo_press = false;
            for (size_t k = 2; k < w.size(); k++) {
                if (w[k] == "pe") D.thermo_pe = true;
                else if (w[k] == "press") D.thermo_press = true;
            }
        } else if (c == "thermo") {
            D.thermo_every = w.size() == 2 ? atoi(w[1].c_str()) : 0;
        } else if (c == "thermo_modify") {
        } else if (c == "timestep") {
            if (w.size() != 2 || !(atof(w[1].c_str()) > 0.0)) { E.err = "Illegal timestep command"; return 1; }
            E.dt = atof(w[1].c_str());
        } else if (c == "run") {
            if (w.size() != 2) { E.err = "Illegal run command"; return 1; }
            int nsteps = atoi(w[1].c_str());
            if ((rc = upload(E, D))) return rc;
            double skin = D.vars.count("__skin") ? atof(D.vars["__skin"].c_str()) : 0.3;
            int every = D.vars.count("__every") ? atoi(D.vars["__every"].c_str()) : 1;
            int delay = D.vars.count("__delay") ? atoi(D.vars["__delay"].c_str()) : 10;
            int chk = D.vars.count("__check") ? (D.vars["__check"] == "yes") : 1;
            if ((rc = E.neighbor(skin, every, delay, chk))) return rc;
            if (!D.is_setup) { if ((rc = E.setup())) return rc; D.is_setup = true; }
            out += "Step Temp CPU S/CPU";
            if (D.thermo_pe) out += " PotEng";
            if (D.thermo_press) out += " Press";
            out += "\n";
            t_run0 = now();
            D.cpu_prev = 0.0;
            D.step_prev = E.ntimestep;
            auto thermo_line = [&]() -> int {
                double T = 0.0;
                int r2 = E.compute_temp(&T);
                if (r2) return r2;
                double cpu = now() - t_run0;
                double spcpu = (cpu > D.cpu_prev) ? (E.ntimestep - D.step_prev) / (cpu - D.cpu_prev) : 0.0;
                snprintf(buf, sizeof buf, "%8ld %12.8g %12.6g %12.6g", (long)E.ntimestep, T, cpu, spcpu);
                out += buf;
                if (D.thermo_pe || D.thermo_press) {
                    // energy / virial tallies of the current configuration (LAMMPS tallies them inside the force evaluation
                    // of a thermo step; here the forces are left untouched so that the trajectory does not depend on the output)
                    if ((r2 = E.tally_ev())) return r2;
                    if (D.thermo_pe) {
                        double pe = 0.0, eb = 0.0, ea = 0.0;
                        if ((r2 = E.compute_pe(&pe)) || (r2 = E.compute_ebond(&eb)) || (r2 = E.compute_eangle(&ea))) return r2;
                        snprintf(buf, sizeof buf, " %14.10g", (pe + eb + ea) / std::max(1, D.natoms));     // thermo_modify norm yes (lj)
                        out += buf;
                    }
                    if (D.thermo_press) {
                        double pr = 0.0;
                        if ((r2 = E.compute_pressure(&pr))) return r2;
                        snprintf(buf, sizeof buf, " %14.10g", pr);
                        out += buf;
                    }
                }
                D.cpu_prev = cpu;
                D.step_prev = E.ntimestep;
                out += "\n";
                return 0;
            };
            if ((rc = thermo_line())) return rc;
            int done = 0;
            while (done < nsteps) {
                int chunk = nsteps - done;
                if (D.thermo_every > 0) {
                    int to_next = D.thermo_every - (int)(E.ntimestep % D.thermo_every);
                    chunk = std::min(chunk, to_next);
                }
                if ((rc = E.run(chunk))) return rc;
                done += chunk;
                if (D.thermo_every > 0 && (E.ntimestep % D.thermo_every == 0 || done == nsteps))
                    if ((rc = thermo_line())) return rc;
            }
            if ((rc = E.sync())) return rc;
            double loop = now() - t_run0;
            snprintf(buf, sizeof buf, "Loop time of %g on 1 procs for %d steps with %d atoms\n", loop, nsteps, D.natoms);
            out += buf;
        } else {
            E.err = "Unknown command: " + line;
            return 1;
        }
    }
    return 0;
}

} // namespace meso
