"""Join of the Hu and PCA tables on `id` (reference data/merge_features.py).

Same behaviour as the reference driver: the id column is found under several
names and renamed to "id" (:6-14), file extensions are stripped from ids
(:16-19), the tables are outer-joined with suffix "_p" for clashes (:45-47),
every other column is coerced to numeric (text becomes NaN, :50-53) and NaN is
filled with 0 (:55).  `merge_frames` is the in-memory form used by the batch
driver after the per-rank feature blocks have been gathered.
"""
from __future__ import annotations

import argparse
from pathlib import Path

import pandas as pd

ID_CANDIDATES = ("ID", "Id", "id", "sample_id", "sample", "filename")


def norm_id_col(df: pd.DataFrame) -> pd.DataFrame:
    for cand in ID_CANDIDATES:
        if cand in df.columns:
            return df.rename(columns={cand: "id"})
    return df.rename(columns={df.columns[0]: "id"})


def strip_ext_from_id(df: pd.DataFrame, id_col: str = "id") -> pd.DataFrame:
    df[id_col] = df[id_col].astype(str).apply(lambda x: Path(x).stem)
    return df


def merge_frames(frames) -> pd.DataFrame:
    """Outer join of string-typed tables on id, numeric coercion, NaN -> 0."""
    frames = [strip_ext_from_id(norm_id_col(f.astype(str))) for f in frames]
    if not frames:
        raise SystemExit("No input CSVs found. Provide at least one of --hu or --pca")
    merged = frames[0]
    for other in frames[1:]:
        merged = merged.merge(other, on="id", how="outer", suffixes=("", "_p"))
    for col in merged.columns:
        if col != "id":
            merged[col] = pd.to_numeric(merged[col], errors="coerce")
    return merged.fillna(0)


def main(args) -> None:
    hu = Path(args.hu_csv) if args.hu_csv else None
    pca = Path(args.pca_csv) if args.pca_csv else None
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    frames = []
    if hu and hu.exists():
        frames.append(pd.read_csv(str(hu), dtype=str))
        print(f"Loaded HU: {hu} ({len(frames[-1])} rows)")
    if pca and pca.exists():
        frames.append(pd.read_csv(str(pca), dtype=str))
        print(f"Loaded PCA: {pca} ({len(frames[-1])} rows)")
    if not frames:
        raise SystemExit("No input CSVs found. Provide at least one of --hu or --pca")
    merged = merge_frames(frames)
    merged.to_csv(str(out), index=False)
    print(f"Saved merged features to {out} ({len(merged)} rows, {len(merged.columns) - 1} features)")


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--hu_csv", type=str, default="src/data/processed/2d_hu/hu_features_table.csv")
    p.add_argument("--pca_csv", type=str, default="src/data/processed/descriptors/descriptors_table_pca.csv")
    p.add_argument("--out", type=str, default="src/data/features/features_table.csv")
    main(p.parse_args())
