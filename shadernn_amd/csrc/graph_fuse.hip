// graph_fuse.hip -- snnhip_graph_fuse: the single graph walk that looks for fusable operator groups (include/snnhip.h).
//
// The fusion RULES live in the chain planner (make_chain_plan, espcn_fused.hip: A/B/C ESPCN kernels, D [UpSampling2D ->] Pad -> Conv2D,
// E Conv2D -> Add, F Conv2D -> InstanceNorm).  This file only decides which groups of a DAG are offered to it:
//   0. inverted-residual blocks with a skip connection: Add(project(depthwise(expand(X))), X) (MobileNetV2);
//   1. residual pairs: an Add one of whose inputs is a convolution that nobody else reads (ResNet skip connections);
//   2. maximal linear runs: node t+1 reads only node t, node t is read only by node t+1 (Candy's pad -> conv -> norm strings, ESPCN).
// Counterpart in the reference: none -- it dispatches one compute shader per layer (vulkanRenderpass.cpp:257-259).
#include <vector>

#include "snnhip_internal.h"

using namespace snnhip;

extern "C" int snnhip_graph_fuse(snnhip_ctx* ctx, const snnhip_graph_node* nodes, int n, snnhip_fused_node* out) {
    SNNHIP_REQUIRE(ctx && nodes && out && n > 0, "graph_fuse: bad argument");
    std::vector<int> consumers(static_cast<size_t>(n), 0);
    for (int i = 0; i < n; ++i) {
        SNNHIP_REQUIRE(nodes[i].n_inputs >= 0 && nodes[i].n_inputs <= SNNHIP_GRAPH_MAX_INPUTS, "graph_fuse: node %d has %d inputs", i, nodes[i].n_inputs);
        for (int k = 0; k < nodes[i].n_inputs; ++k) {
            const int src = nodes[i].inputs[k];
            SNNHIP_REQUIRE(src < i, "graph_fuse: node %d reads node %d (nodes must be in execution order)", i, src);
            if (src >= 0) consumers[static_cast<size_t>(src)]++;
        }
        out[i].plan = nodes[i].plan;
        out[i].owned = 0;
        out[i].n_inputs = nodes[i].n_inputs;
        for (int k = 0; k < SNNHIP_GRAPH_MAX_INPUTS; ++k) out[i].inputs[k] = k < nodes[i].n_inputs ? nodes[i].inputs[k] : 0;
    }
    SNNHIP_CHECK_HIP(hipSetDevice(ctx->device));
    std::vector<char> taken(static_cast<size_t>(n), 0); // member of a fused group already
    auto foldable = [&](int i) { return i >= 0 && nodes[i].plan && !nodes[i].keep && consumers[static_cast<size_t>(i)] == 1 && !taken[static_cast<size_t>(i)]; };
    auto fail = [&](int rc) { // hand back nothing half-built
        for (int i = 0; i < n; ++i)
            if (out[i].owned && out[i].plan) {
                delete out[i].plan;
                out[i].plan = nodes[i].plan;
                out[i].owned = 0;
            }
        return rc;
    };

    // ---- 0. inverted-residual block with skip connection (rule G): X -> Conv2D 1x1 -> DepthwiseConv2D -> Conv2D 1x1 -> Add(., X) -> one plan at
    // the Add node that reads X only
    for (int r = 0; r < n; ++r) {
        if (!nodes[r].plan || nodes[r].n_inputs != 2 || taken[static_cast<size_t>(r)]) continue;
        auto* add = dynamic_cast<EltwisePlanBase*>(nodes[r].plan);
        if (!add || add->mode != 0) continue;
        for (int which = 0; which < 2; ++which) {
            const int pj = nodes[r].inputs[which];
            if (!foldable(pj) || nodes[pj].n_inputs != 1) continue;
            const int dw = nodes[pj].inputs[0];
            if (!foldable(dw) || nodes[dw].n_inputs != 1) continue;
            const int ex = nodes[dw].inputs[0];
            if (!foldable(ex) || nodes[ex].n_inputs != 1 || nodes[ex].inputs[0] != nodes[r].inputs[1 - which]) continue;
            snnhip_plan* fused = nullptr;
            const int rc = make_irb_plan(ctx, nodes[ex].plan, nodes[dw].plan, nodes[pj].plan, nodes[r].plan, &fused);
            if (rc == SNNHIP_E_UNSUPPORTED) continue;
            if (rc != SNNHIP_OK) return fail(rc);
            out[r].plan = fused;
            out[r].owned = 1;
            out[r].n_inputs = 1;
            out[r].inputs[0] = nodes[ex].inputs[0];
            for (int t : {ex, dw, pj}) {
                out[t].plan = nullptr;
                out[t].n_inputs = 0;
                taken[static_cast<size_t>(t)] = 1;
            }
            taken[static_cast<size_t>(r)] = 1;
            break;
        }
    }

    // ---- 0b. Flatten (an Activation with no activation) -> Dense: flattening contiguous NHWC memory in HWC order is the identity, and the dense plan
    // takes any [N,H,W,C] tensor of the right element count -- it reads the Flatten's producer, the copy launch disappears
    for (int dn = 0; dn < n; ++dn) {
        if (!nodes[dn].plan || nodes[dn].n_inputs != 1 || taken[static_cast<size_t>(dn)] || nodes[dn].plan->desc.rfind("dense", 0) != 0) continue;
        const int fl = nodes[dn].inputs[0];
        if (!foldable(fl) || nodes[fl].n_inputs != 1) continue;
        auto* id = dynamic_cast<EltwisePlanBase*>(nodes[fl].plan);
        if (!id || id->mode != 1 || id->d.act != SNNHIP_ACT_NONE) continue;
        out[dn].inputs[0] = nodes[fl].inputs[0];
        out[fl].plan = nullptr;
        out[fl].n_inputs = 0;
        taken[static_cast<size_t>(fl)] = 1; // the Dense node itself stays available for other rules (none applies today)
    }

    // ---- 1. Conv2D -> Add (rule E): the fused plan sits at the Add node and reads {the convolution's input, the other summand}
    for (int k = 0; k < n; ++k) {
        if (!nodes[k].plan || nodes[k].n_inputs != 2 || taken[static_cast<size_t>(k)]) continue;
        auto* add = dynamic_cast<EltwisePlanBase*>(nodes[k].plan);
        if (!add || add->mode != 0) continue;
        for (int which = 0; which < 2; ++which) {
            const int src = nodes[k].inputs[which];
            if (!foldable(src) || nodes[src].n_inputs != 1) continue;
            auto* cv = dynamic_cast<ConvPlanBase*>(nodes[src].plan);
            if (!cv || cv->depthwise) continue;
            if (nodes[k].inputs[1 - which] == src) continue; // x + x
            snnhip_plan* pair[2] = {nodes[src].plan, nodes[k].plan};
            snnhip_plan* fused = nullptr;
            const int rc = make_chain_plan(ctx, pair, 2, &fused);
            if (rc == SNNHIP_E_UNSUPPORTED) continue;
            if (rc != SNNHIP_OK) return fail(rc);
            if (fused->numInputs != 2) { // the chain planner wrapped the pair without folding the add: not what this rule is for
                delete fused;
                continue;
            }
            out[k].plan = fused;
            out[k].owned = 1;
            out[k].n_inputs = 2;
            out[k].inputs[0] = nodes[src].inputs[0];
            out[k].inputs[1] = nodes[k].inputs[1 - which];
            out[src].plan = nullptr;
            out[src].n_inputs = 0;
            taken[static_cast<size_t>(k)] = taken[static_cast<size_t>(src)] = 1;
            break;
        }
    }

    // ---- 1b. InstanceNorm -> Add (rule H: the residual blocks of the style-transfer networks): the Add moves into the norm's normalise sweep
    for (int k = 0; k < n; ++k) {
        if (!nodes[k].plan || nodes[k].n_inputs != 2 || taken[static_cast<size_t>(k)]) continue;
        auto* add = dynamic_cast<EltwisePlanBase*>(nodes[k].plan);
        if (!add || add->mode != 0) continue;
        for (int which = 0; which < 2; ++which) {
            const int src = nodes[k].inputs[which];
            if (!foldable(src) || nodes[src].n_inputs != 1 || !instancenorm_plan_desc(nodes[src].plan, nullptr)) continue;
            if (nodes[k].inputs[1 - which] == src) continue;
            snnhip_plan* fused = nullptr;
            const int rc = make_instancenorm_add_plan(ctx, nodes[src].plan, nodes[k].plan, which == 0, &fused);
            if (rc == SNNHIP_E_UNSUPPORTED) continue;
            if (rc != SNNHIP_OK) return fail(rc);
            out[k].plan = fused;
            out[k].owned = 1;
            out[k].n_inputs = 2;
            out[k].inputs[0] = nodes[src].inputs[0];
            out[k].inputs[1] = nodes[k].inputs[1 - which];
            out[src].plan = nullptr;
            out[src].n_inputs = 0;
            taken[static_cast<size_t>(k)] = taken[static_cast<size_t>(src)] = 1;
            break;
        }
    }

    // ---- 2. maximal linear runs -> chain planner
    for (int i = 0; i < n;) {
        if (!nodes[i].plan || nodes[i].n_inputs != 1 || taken[static_cast<size_t>(i)]) {
            ++i;
            continue;
        }
        int j = i;
        while (j + 1 < n && nodes[j + 1].plan && nodes[j + 1].n_inputs == 1 && nodes[j + 1].inputs[0] == j && foldable(j) && !taken[static_cast<size_t>(j + 1)]) ++j;
        // the run may end in front of an InstanceNorm -> Add node built in 1b (a residual block's tail): offered to the planner as the chain's
        // two-input last step, so that the convolution in front can hand its tile statistics to the norm (rules F + H)
        int tail = -1;
        if (foldable(j))
            for (int k = j + 1; k < n && tail < 0; ++k)
                if (out[k].owned && out[k].plan && out[k].n_inputs == 2 && out[k].inputs[0] == j && out[k].inputs[1] != j && out[k].plan->numInputs == 2 &&
                    instancenorm_add_use_tile_stats(out[k].plan, TileStatsRef()))
                    tail = k;
        if (tail >= 0) {
            std::vector<snnhip_plan*> run;
            for (int t = i; t <= j; ++t) run.push_back(nodes[t].plan);
            run.push_back(out[tail].plan);
            snnhip_plan* chain = nullptr;
            const int rc = make_chain_plan(ctx, run.data(), static_cast<int>(run.size()), &chain);
            if (rc == SNNHIP_OK && chain_adopt_plan(chain, out[tail].plan)) {
                out[tail].plan = chain;
                out[tail].inputs[0] = out[i].n_inputs > 0 ? out[i].inputs[0] : nodes[i].inputs[0]; // the CURRENT wiring (rule 0b re-wires a Dense past its Flatten)
                for (int t = i; t <= j; ++t) {
                    out[t].plan = nullptr;
                    out[t].n_inputs = 0;
                    taken[static_cast<size_t>(t)] = 1;
                }
                i = j + 1;
                continue;
            }
            if (rc == SNNHIP_OK) delete chain; // (a plan that is not a chain: no rule of this walk builds one here)
            else if (rc != SNNHIP_E_UNSUPPORTED) return fail(rc);
        }
        if (j > i) {
            std::vector<snnhip_plan*> run;
            for (int t = i; t <= j; ++t) run.push_back(nodes[t].plan);
            snnhip_plan* chain = nullptr;
            const int rc = make_chain_plan(ctx, run.data(), static_cast<int>(run.size()), &chain);
            if (rc == SNNHIP_OK) {
                const int firstInput = out[i].n_inputs > 0 ? out[i].inputs[0] : nodes[i].inputs[0]; // the CURRENT wiring, read before out[i] is cleared
                out[j].plan = chain;
                out[j].owned = 1;
                out[j].n_inputs = 1;
                out[j].inputs[0] = firstInput;
                for (int t = i; t < j; ++t) {
                    out[t].plan = nullptr;
                    out[t].n_inputs = 0;
                }
                for (int t = i; t <= j; ++t) taken[static_cast<size_t>(t)] = 1;
            } else if (rc != SNNHIP_E_UNSUPPORTED) {
                return fail(rc);
            }
        }
        i = j + 1;
    }
    return SNNHIP_OK;
}
