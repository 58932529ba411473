"""Material dictionaries with the names MoonRTX imports from plotoptix.materials.

Only the keys the reference reads or writes are meaningful here: `m_diffuse.copy()` + "ColorTextures"
(moon_renderer.py:615-616) and `m_flat` + "OcclusionProgram" / "VarFloat4" (renderer_labels.py:132-139).
The programs named in the values are PlotOptiX's PTX entry points; this backend selects its shading by
material *kind* ("diffuse" = Lambert with optional colour texture, "flat" = emissive, never shadows).
"""
m_diffuse = {
    "kind": "diffuse",
    "ClosestHitPrograms": ["0::path_tracing_materials.ptx::__closesthit__diffuse"],
    "AnyHitPrograms": ["1::path_tracing_materials.ptx::__anyhit__occlusion"],
    "VarUInt": {"flags": 2},
    "ColorTextures": [],
}

m_flat = {
    "kind": "flat",
    "ClosestHitPrograms": ["0::path_tracing_materials.ptx::__closesthit__flat"],
    "AnyHitPrograms": ["1::path_tracing_materials.ptx::__anyhit__occlusion"],
    "VarUInt": {"flags": 1},
}
