#!/usr/bin/env python3
"""Extract the numeric parameters of the reference's 'thin' obstacle set into a data table.

    python tools/extract_thin_assets.py     (build container only; needs /root/reference)

Each airgym/assets/env_assets/thin/tree_<k>.urdf holds ONE tilted cylinder: radius, length, origin xyz, rpy
(SURVEY 8(f)-1).  Only these eight numbers per variant are written (airgym_amd/assets/thin_trees.json);
no URDF text is copied.
"""
import json
import os
import re
import xml.etree.ElementTree as ET

SRC = "/root/reference/airgym/assets/env_assets/thin"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "airgym_amd", "assets", "thin_trees.json")

rows = []
for k in range(100):
    root = ET.parse(os.path.join(SRC, f"tree_{k}.urdf")).getroot()
    col = root.find("link").find("collision")
    cyl = col.find("geometry").find("cylinder")
    org = col.find("origin")
    xyz = [float(x) for x in org.get("xyz").split()]
    rpy = [float(x) for x in org.get("rpy").split()]
    rows.append([float(cyl.get("radius")), float(cyl.get("length"))] + xyz + rpy)
json.dump({"source": "emNavi/AirGym airgym/assets/env_assets/thin/tree_<k>.urdf, k = 0..99 (collision cylinder)",
           "columns": ["radius", "length", "ox", "oy", "oz", "roll", "pitch", "yaw"], "variants": rows},
          open(DST, "w"), indent=0)
print("wrote", DST, len(rows))
