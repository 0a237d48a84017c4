// FrequenSee.hpp — C++17 host-side mirror of the reference's plugin interface, over the C ABI only
// (include/frequensee.h; no HIP headers needed by the includer).
//
//   frequensee::AudioRayTracingSubsystem  <->  UAudioRayTracingSubsystem  (Public/AudioRayTracingSubsystem.h:86-196)
//   frequensee::FrequenSeeAudioComponent  <->  UFrequenSeeAudioComponent  (Public/FrequenSeeAudioComponent.h:20-154)
//   frequensee::MaterialAcousticProcessor <->  UMaterialAcousticProcessor (Public/MaterialAcousticProcessor.h:55-70)
//
// Same member names and argument meaning as the reference so a UE shim or a headless
// harness reads like the original call sites.  check()-style aborts of the reference become
// std::runtime_error carrying fs_last_error().
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/frequensee.h"

namespace frequensee {

struct FVector {
    float X = 0, Y = 0, Z = 0;
};

class AudioRayTracingSubsystem;
class MaterialAcousticProcessor;

// UAcousticGeometryComponent (Public/AcousticGeometryComponent.h:9-22) + the owner's collision triangles
struct AcousticGeometryComponent {
    std::vector<float> Triangles;     // [T][3][3], cm
    std::vector<uint16_t> MaterialId; // [T] index into the material table (UAcousticMaterial per actor)
    uint32_t Actor = 0;               // AActor the collision belongs to
};

class FrequenSeeAudioComponent {
public:
    explicit FrequenSeeAudioComponent(FVector Location = {}) : Location_(Location) {}
    FrequenSeeAudioComponent(const FrequenSeeAudioComponent&) = delete;
    FrequenSeeAudioComponent& operator=(const FrequenSeeAudioComponent&) = delete;
    ~FrequenSeeAudioComponent();

    void OnRegister(AudioRayTracingSubsystem& SubSys);   // FrequenSeeAudioComponent.cpp:42-52
    void OnUnregister();                                 // :54-64

    FVector GetComponentLocation() const { return Location_; }
    void SetComponentLocation(FVector L);

    int NumBins() const;      // FrequenSeeAudioComponent.h:137
    int NumSamples() const;   // :138

    void FlushEnergyBuffer();                                            // .h:76-79
    void UpdateEnergyBuffer(const std::vector<float>& NewEnergyValues);  // .h:81-85
    void AddEnergyAtDelay(float DelaySeconds, float EnergyValue, int Band = 0);   // .h:87-91
    std::vector<float> EnergyBuffer() const;                             // [bands][NumBins]
    void ReconstructImpulseResponse();                                   // .cpp:320-380
    // GetImpulseResponse()[Channel]: pointer into the published front buffer (lock-free, audio thread safe)
    const float* GetImpulseResponse(int Channel, int* NumSamplesOut = nullptr) const;   // .h:113
    float GetOcclusionAttenuation() const;                               // .h:112
    fs_sound_result UpdateSound(uint64_t Seed = 0x5EED);                 // .cpp:283-306
    void SaveImpulseResponse(const std::string& Path, int Channel = 0) const;   // SaveArrayToFile .cpp:492-505
    // GetImpulseResponse() is a mutable reference in the reference (.h:113): install an IR of the caller's own
    void SetImpulseResponse(const std::vector<float>& IR);
    // GenerateDummyImpulseResponse (.cpp:408-452) as it ends up: a delta at samples 0 and N-1
    std::vector<float> GenerateDummyImpulseResponse();

    bool bApplyReverb = true;   // .h:60

private:
    friend class AudioRayTracingSubsystem;
    FVector Location_;
    AudioRayTracingSubsystem* SubSys_ = nullptr;
    fs_source Handle_ = -1;
};

class AudioRayTracingSubsystem {
public:
    static constexpr int USED_RAY_COUNT = 1000;   // AudioRayTracingSubsystem.h:176

    explicit AudioRayTracingSubsystem(int NumBands = 1, int Device = 0, int Rank = 0, int WorldSize = 1) {   // Initialize :32-36
        fs_config c;
        fs_config_default(&c);
        c.num_bands = NumBands; c.device = Device; c.rank = Rank; c.world_size = WorldSize;
        NumBands_ = NumBands;
        int rc = fs_context_create(&c, &Ctx_);
        if (rc != FS_OK) {
            std::string m = Ctx_ ? fs_last_error(Ctx_) : "fs_context_create failed";
            if (Ctx_) fs_context_destroy(Ctx_);
            Ctx_ = nullptr;
            throw std::runtime_error("FrequenSee: " + m);
        }
        fs_params_default(&Params);
        Params.num_rays = 2 * USED_RAY_COUNT;
    }
    ~AudioRayTracingSubsystem() { if (Ctx_) fs_context_destroy(Ctx_); }   // Deinitialize :38-42
    AudioRayTracingSubsystem(const AudioRayTracingSubsystem&) = delete;
    AudioRayTracingSubsystem& operator=(const AudioRayTracingSubsystem&) = delete;

    // RegisterGeometry / UnregisterGeometry (.h:99-100).  The first commit builds the tree on the host (binned SAH);
    // a registration change DURING play (Committed_ already) rebuilds it on the device instead, a tenth of the time
    // (fs_scene_commit_progressive: the SAH tree follows by itself from a background thread; RebuildQuality() forces it now).
    void RegisterGeometry(const AcousticGeometryComponent* Comp) { Geometry_.push_back(Comp); Dirty_ = true; }
    void UnregisterGeometry(const AcousticGeometryComponent* Comp) {
        for (size_t i = 0; i < Geometry_.size(); ++i)
            if (Geometry_[i] == Comp) { Geometry_.erase(Geometry_.begin() + (long)i); Dirty_ = true; break; }
    }
    void RebuildQuality() { Dirty_ = true; Committed_ = false; Commit(); }
    // Multi-GPU (one process per GPU, Rank / WorldSize given to the constructor): Id = the FS_COMM_ID_BYTES bytes rank 0
    // got from fs_comm_unique_id, shipped to every rank.  From then on every frame's energy buffer is summed over the
    // ranks inside the library and Commit() lets rank 0 build the tree for all.
    void CommInit(const void* Id) { Check(fs_comm_init(Ctx_, Id, FS_COMM_ID_BYTES)); }
    // One source per GPU (cfg5): no sharding, no reduce — a peer communicator only for GatherEnergy, after which any
    // rank can install a peer's histogram (UpdateEnergyBuffer on a mirror component) and serve its IR.
    void PeersInit(const void* Id, int Rank, int WorldSize) { Check(fs_peers_init(Ctx_, Id, FS_COMM_ID_BYTES, Rank, WorldSize)); Peers_ = WorldSize; }
    std::vector<float> GatherEnergy(fs_source Src) {             // [WorldSize][bands][bins], collective
        std::vector<float> Out((size_t)Peers_ * (size_t)NumBands_ * (size_t)fs_num_bins(Ctx_));
        Check(fs_gather_energy(Ctx_, Src, Out.data(), (int32_t)Out.size()));
        return Out;
    }
    // UAcousticMaterial table: Absorption [M][bands] (AcousticMaterial.h:22-30)
    void SetMaterials(const std::vector<float>& Absorption, int NumMaterials) {
        Absorption_ = Absorption; NumMaterials_ = NumMaterials; Dirty_ = true;
    }
    void RegisterSource(FrequenSeeAudioComponent* InComp) {      // .cpp:45-48
        Check(fs_source_create(Ctx_, &InComp->Handle_));
        InComp->SubSys_ = this;
        const FVector L = InComp->Location_;
        Check(fs_source_set_position(Ctx_, InComp->Handle_, &L.X));
        ActiveSources.push_back(InComp);
    }
    void UnRegisterSource(FrequenSeeAudioComponent* InComp) {    // .cpp:50-53
        for (size_t i = 0; i < ActiveSources.size(); ++i)
            if (ActiveSources[i] == InComp) {
                ActiveSources.erase(ActiveSources.begin() + (long)i);
                fs_source_destroy(Ctx_, InComp->Handle_);
                InComp->Handle_ = -1; InComp->SubSys_ = nullptr;
                break;
            }
    }
    void SetListenerLocation(FVector L) { Check(fs_listener_set_position(Ctx_, &L.X)); }   // PlayerPawn location :287

    // UpdateSource (.cpp:128-195): trace + evaluate + flush + deposit + reconstruct
    void UpdateSource(FrequenSeeAudioComponent& Src, std::vector<float>* EnergyOut = nullptr) {
        Commit();
        const int n = Src.NumBins() * NumBands();
        if (EnergyOut) EnergyOut->resize((size_t)n);
        Check(fs_compute_energy_response(Ctx_, Src.Handle_, &Params, EnergyOut ? EnergyOut->data() : nullptr));
        Check(fs_reconstruct_impulse_response(Ctx_, Src.Handle_, &Params));
    }
    // .cpp:883-886 — every active source; one batched frame on the device (fs_compute_energy_response_batch_async)
    void ForceUpdateSources() {
        if (ActiveSources.empty()) return;
        Commit();
        std::vector<fs_source> H;
        for (auto* s : ActiveSources) H.push_back(s->Handle_);
        Check(fs_update_sources(Ctx_, H.data(), (int32_t)H.size(), &Params));   // one batched frame, one reconstruct launch, one wait
    }
    // .cpp:55-85 (the caller drives every frame).  The reference draws from the engine's global rand() stream, so every
    // frame sees fresh samples: the seed advances.
    void Tick(float /*DeltaTime*/) {
        if (ActiveSources.empty()) return;
        if (Streamed_) {   // the reference's loop over the sources (.cpp:60-68), one pipelined launch per source
            Commit();
            for (auto* s : ActiveSources) {
                Check(fs_compute_energy_response_async(Ctx_, s->Handle_, &Params));
                Check(fs_reconstruct_impulse_response_async(Ctx_, s->Handle_, &Params));
            }
            Check(fs_submit(Ctx_));   // nothing waits: the IRs appear as they are published
        } else {
            ForceUpdateSources();
        }
        ++Params.seed;
    }
    // fs_set_pipelining (0 off, 1, 2): Tick streams the sources instead of batching them
    void SetPipelining(int Depth) { Check(fs_set_pipelining(Ctx_, Depth)); Streamed_ = Depth != 0; }
    // fs_set_frames_per_launch (1 .. 4): consecutive streamed frames share a launch (each keeps its seed, buffer and IR)
    void SetFramesPerLaunch(int N) { Check(fs_set_frames_per_launch(Ctx_, N)); }
    void Synchronize() { Check(fs_synchronize(Ctx_)); }

    int NumBands() const { return NumBands_; }
    fs_context* Context() const { return Ctx_; }
    void Commit() {
        if (!Dirty_) return;
        std::vector<float> xyz; std::vector<uint16_t> mat; std::vector<uint32_t> obj;
        for (const auto* g : Geometry_) {
            xyz.insert(xyz.end(), g->Triangles.begin(), g->Triangles.end());
            mat.insert(mat.end(), g->MaterialId.begin(), g->MaterialId.end());
            obj.insert(obj.end(), g->MaterialId.size(), g->Actor);
        }
        Check(fs_scene_set_triangles(Ctx_, xyz.data(), mat.data(), (int32_t)mat.size()));
        Check(fs_scene_set_materials(Ctx_, Absorption_.data(), nullptr, nullptr, NumMaterials_,
                                     NumMaterials_ ? (int32_t)(Absorption_.size() / (size_t)NumMaterials_) : NumBands()));
        Check(fs_scene_set_objects(Ctx_, obj.data(), (int32_t)obj.size()));
        Check(Committed_ ? fs_scene_commit_progressive(Ctx_) : fs_scene_commit(Ctx_));
        Dirty_ = false;
        Committed_ = true;
    }
    // A registered geometry component moved (ECC_WorldDynamic prop): its triangles are rewritten in place and the
    // acceleration structure is refitted on the device before the next trace — no rebuild.  `Comp->Triangles`
    // already holds the new world-space positions; the triangle count must not change.
    void GeometryMoved(const AcousticGeometryComponent* Comp) {
        Commit();
        size_t first = 0;
        for (const auto* g : Geometry_) {
            if (g == Comp) {
                Check(fs_scene_update_triangles(Ctx_, (int32_t)first, (int32_t)g->MaterialId.size(), g->Triangles.data()));
                return;
            }
            first += g->MaterialId.size();
        }
        throw std::runtime_error("FrequenSee: geometry component is not registered");
    }
    void Check(int rc) const { if (rc != FS_OK) throw std::runtime_error(std::string("FrequenSee: ") + fs_last_error(Ctx_)); }

    fs_params Params;                                   // the constants of AudioRayTracingSubsystem.cpp:282-284, 362-413
    std::vector<FrequenSeeAudioComponent*> ActiveSources;

private:
    fs_context* Ctx_ = nullptr;
    std::vector<const AcousticGeometryComponent*> Geometry_;
    std::vector<float> Absorption_;
    int NumMaterials_ = 0;
    bool Dirty_ = true;
    bool Committed_ = false;
    int NumBands_ = 1;
    int Peers_ = 0;
    bool Streamed_ = false;
    friend class FrequenSeeAudioComponent;
    friend class MaterialAcousticProcessor;
};

inline FrequenSeeAudioComponent::~FrequenSeeAudioComponent() { OnUnregister(); }
inline void FrequenSeeAudioComponent::OnRegister(AudioRayTracingSubsystem& S) { S.RegisterSource(this); }
inline void FrequenSeeAudioComponent::OnUnregister() { if (SubSys_) SubSys_->UnRegisterSource(this); }
inline void FrequenSeeAudioComponent::SetComponentLocation(FVector L) {
    Location_ = L;
    if (SubSys_) SubSys_->Check(fs_source_set_position(SubSys_->Ctx_, Handle_, &L.X));
}
inline int FrequenSeeAudioComponent::NumBins() const { return fs_num_bins(SubSys_->Ctx_); }
inline int FrequenSeeAudioComponent::NumSamples() const { return fs_num_samples(SubSys_->Ctx_); }
inline void FrequenSeeAudioComponent::FlushEnergyBuffer() { SubSys_->Check(fs_flush_energy_buffer(SubSys_->Ctx_, Handle_)); }
inline void FrequenSeeAudioComponent::UpdateEnergyBuffer(const std::vector<float>& V) {
    SubSys_->Check(fs_update_energy_buffer(SubSys_->Ctx_, Handle_, V.data(), (int32_t)V.size()));
}
inline void FrequenSeeAudioComponent::AddEnergyAtDelay(float D, float E, int Band) {
    SubSys_->Check(fs_add_energy_at_delay(SubSys_->Ctx_, Handle_, Band, D, E));
}
inline std::vector<float> FrequenSeeAudioComponent::EnergyBuffer() const {
    std::vector<float> v((size_t)NumBins() * (size_t)SubSys_->NumBands());
    SubSys_->Check(fs_get_energy_buffer(SubSys_->Ctx_, Handle_, v.data(), (int32_t)v.size()));
    return v;
}
inline void FrequenSeeAudioComponent::ReconstructImpulseResponse() {
    SubSys_->Check(fs_reconstruct_impulse_response(SubSys_->Ctx_, Handle_, &SubSys_->Params));
}
inline const float* FrequenSeeAudioComponent::GetImpulseResponse(int Channel, int* N) const {
    const float* p = nullptr;
    int32_t n = 0;
    SubSys_->Check(fs_get_impulse_response(SubSys_->Ctx_, Handle_, Channel, &p, &n));
    if (N) *N = n;
    return p;
}
inline float FrequenSeeAudioComponent::GetOcclusionAttenuation() const {
    float v = 1.f;
    SubSys_->Check(fs_get_occlusion_attenuation(SubSys_->Ctx_, Handle_, &v));
    return v;
}
inline fs_sound_result FrequenSeeAudioComponent::UpdateSound(uint64_t Seed) {
    SubSys_->Commit();
    fs_sound_params p;
    fs_sound_params_default(&p);
    p.seed = Seed;
    fs_sound_result r{};
    SubSys_->Check(fs_update_sound(SubSys_->Ctx_, Handle_, &p, &r));
    return r;
}
inline void FrequenSeeAudioComponent::SetImpulseResponse(const std::vector<float>& IR) {
    SubSys_->Check(fs_set_impulse_response(SubSys_->Ctx_, Handle_, IR.data(), (int32_t)IR.size()));
}
inline std::vector<float> FrequenSeeAudioComponent::GenerateDummyImpulseResponse() {
    std::vector<float> IR((size_t)fs_num_samples(SubSys_->Ctx_), 0.0f);
    if (!IR.empty()) { IR.front() = 1.0f; IR.back() = 1.0f; }
    SetImpulseResponse(IR);
    return IR;
}
inline void FrequenSeeAudioComponent::SaveImpulseResponse(const std::string& Path, int Channel) const {
    SubSys_->Check(fs_save_impulse_response(SubSys_->Ctx_, Handle_, Channel, Path.c_str()));
}

// FMaterialAcousticFD / FAcousticOutputs (Public/MaterialAcousticProcessor.h:24-53)
struct MaterialAcousticFD {
    std::vector<float> Absorption, Transmission, Scattering;   // each N/2 + 1 responses
};
struct AcousticOutputs {
    std::vector<float> Specular, Diffuse, Transmitted;
};

class MaterialAcousticProcessor {
public:
    explicit MaterialAcousticProcessor(AudioRayTracingSubsystem& SubSys) : SubSys_(&SubSys) {}
    // ApplyMaterialFD (MaterialAcousticProcessor.cpp:8-107).  A wrong curve length logs an error and returns
    // empty outputs in the reference (:20-26); here it throws with the same message.
    AcousticOutputs ApplyMaterialFD(const std::vector<float>& InBuffer, const MaterialAcousticFD& Props) const {
        AcousticOutputs Out;
        const size_t L = InBuffer.size();
        if (Props.Absorption.size() != Props.Transmission.size() || Props.Absorption.size() != Props.Scattering.size())
            throw std::runtime_error("FrequenSee: response curves differ in length");
        Out.Specular.resize(L); Out.Diffuse.resize(L); Out.Transmitted.resize(L);
        SubSys_->Check(fs_apply_material_fd(SubSys_->Ctx_, InBuffer.data(), (int32_t)L, Props.Absorption.data(),
                                            Props.Transmission.data(), Props.Scattering.data(),
                                            (int32_t)Props.Absorption.size(), Out.Specular.data(), Out.Diffuse.data(),
                                            Out.Transmitted.data()));
        return Out;
    }

private:
    AudioRayTracingSubsystem* SubSys_;
};

}  // namespace frequensee
