"""On-GPU counterpart of the reference's stack-mode collate (geotransformer/utils/data.py:13-97,159-209).

The reference runs grid subsampling and the 3S-2 radius searches on the CPU inside DataLoader workers; here the
whole pyramid is built on the GPU in the main process right after the raw pair was uploaded, with two host
synchronisations (the counts of all subsampling stages; the widths of all neighbour tables + the number of rows with exact distance
ties).  Clouds that hold exact ties (real scans) take one more per support stage: a host copy of the stage's points, from which the
reference's k-d tree is built for the device pass that gives those rows the reference's order (csrc/radius_ties.hip).  List structure,
voxel/radius doubling, the 2000-point cap of the coarsest stage and the column truncation are reproduced exactly
(pinned by tests/golden/precompute_c1.npz)."""
import torch

from . import ops as _ops
from .modules.ops import grid_subsample


def precompute_data_stack_mode(points, lengths, num_stages, voxel_size, radius, neighbor_limits):
    """points (N, 3) float32 GPU tensor (ref rows then src rows; several pairs may be stacked: ref0, src0, ref1, src1, ...),
    lengths (2 B,) int64 (host).  Returns the dict of lists
    {'points', 'lengths', 'neighbors', 'subsampling', 'upsampling'}; `lengths` entries are host int64 tensors."""
    assert num_stages == len(neighbor_limits)
    if not points.is_cuda:
        raise RuntimeError('precompute_data_stack_mode: points must be on the GPU')
    lengths = torch.as_tensor(lengths, dtype=torch.int64).cpu()
    points_list, lengths_list = [], []
    # the S-1 subsampling stages back to back: every stage takes the per-cloud counts of the one before from DEVICE memory (outputs sized
    # by the stage-0 row count, an upper bound), and ONE synchronisation fetches the counts of all stages
    sub_pts, sub_len, cur_pts, cur_len, v = [], [], points, lengths, voxel_size
    for i in range(1, num_stages):
        v *= 2
        cur_pts, _, cur_len = _ops.grid_subsample(cur_pts, cur_len, None, v)
        sub_pts.append(cur_pts)
        sub_len.append(cur_len)
    sub_len = torch.stack(sub_len).cpu() if sub_len else None
    for i in range(num_stages):
        if i > 0:
            lengths = sub_len[i - 1].clone()
            points = sub_pts[i - 1][:int(lengths.sum())]
        if i == num_stages - 1 and int(lengths.max()) > 2000:
            # the reference keeps at most 2000 superpoints per cloud (utils/data.py:40-48); any number of stacked clouds here
            keep, start = [], 0
            for n in lengths.tolist():
                keep.append(points[start:start + min(n, 2000)])
                start += n
            points = torch.cat(keep, 0)
            lengths = torch.clamp(lengths, max=2000)
        points_list.append(points.contiguous())
        lengths_list.append(lengths)
        voxel_size *= 2

    # a spatial order of every stage's points for the union-staged KPConv (tile membership only; csrc/kpconv_union.hip)
    # (the default policy runs that kernel on the layers whose queries are stage 0 / 1 points: ops._kpconv_union_pays)
    # ... and only on stacked batches: with one pair per forward (10 000 stage-0 points) the order and the plans cost what the kernels gain
    ns = num_stages if _ops.KPCONV_UNION_ALL else min(2, num_stages)
    if _ops.KPCONV_UNION and (_ops.KPCONV_UNION_ALL or points_list[0].shape[0] >= _ops.KPCONV_UNION_MIN_POINTS):
        _ops.register_point_orders(points_list[:ns], lengths_list[:ns], [voxel_size / 2 ** (num_stages - i) for i in range(ns)])
    # all 3S-2 searches are launched back to back; their column counts are fetched with ONE synchronisation
    jobs = []
    grids = {}

    def grid_for(stage, r):       # one cell grid per (support stage, radius): shared by up to three searches
        key = (stage, float(r))
        if key not in grids:
            s_pts = points_list[stage]
            grids[key] = (_ops.RadiusGrid(s_pts, lengths_list[stage], r)
                          if s_pts.shape[0] >= _ops.GRID_SEARCH_MIN_SUPPORT else None)
        return grids[key]

    n_jobs, n_clouds = 3 * num_stages - 2, len(lengths_list[0])
    # one fill for the counters of all searches: per job the largest in-radius count of every cloud, then per job the number of rows whose
    # kept columns hold an EXACT distance tie (the reference orders those by its k-d tree walk: csrc/radius_ties.hip)
    words = torch.zeros((n_jobs * n_clouds + n_jobs,), dtype=torch.int32, device=points.device)
    counters, tie_counts = words[:n_jobs * n_clouds].view(n_jobs, n_clouds), words[n_jobs * n_clouds:]
    want_ties = _ops.RADIUS_REFERENCE_TIES
    q_rows = [points_list[i].shape[0] for i in range(num_stages)]
    tie_rows = torch.empty((sum(q_rows) + sum(q_rows[1:]) + sum(q_rows[:-1]),), dtype=torch.int32, device=points.device) if want_ties else None
    tie_at = [0]

    def ties_for(nq):             # this job's slice of the row list + its counter word
        if not want_ties:
            return None
        j, o = len(jobs), tie_at[0]
        tie_at[0] += nq
        return tie_rows[o:o + nq], tie_counts[j:j + 1]

    searches = []                 # per job: (queries, support, q_lengths, s_lengths, radius, support stage, ties)
    for i in range(num_stages):
        cur, cl = points_list[i], lengths_list[i]
        t = ties_for(cur.shape[0])
        searches.append((cur, cur, cl, cl, radius, i, t))
        jobs.append(('neighbors', _ops.radius_neighbors(cur, cur, cl, cl, radius, neighbor_limits[i], grid=grid_for(i, radius),
                                                        zeroed_max_count=counters[len(jobs)], ties=t)))
        if i < num_stages - 1:
            sub, sl = points_list[i + 1], lengths_list[i + 1]
            t = ties_for(sub.shape[0])
            searches.append((sub, cur, sl, cl, radius, i, t))
            jobs.append(('subsampling', _ops.radius_neighbors(sub, cur, sl, cl, radius, neighbor_limits[i],
                                                              grid=grid_for(i, radius), zeroed_max_count=counters[len(jobs)], ties=t)))
            t = ties_for(cur.shape[0])
            searches.append((cur, sub, cl, sl, radius * 2, i + 1, t))
            jobs.append(('upsampling', _ops.radius_neighbors(cur, sub, cl, sl, radius * 2, neighbor_limits[i + 1],
                                                             grid=grid_for(i + 1, radius * 2), zeroed_max_count=counters[len(jobs)], ties=t)))
        radius *= 2
    for j, (_, (_, mc)) in enumerate(jobs):       # (a search on a small support takes the exhaustive kernel, which fills a counter of its own)
        if mc.data_ptr() != counters[j].data_ptr():
            counters[j].copy_(mc)
    words_host = words.cpu()                                                      # the ONE synchronisation of the searches
    counts = words_host[:n_jobs * n_clouds].view(n_jobs, n_clouds)                # (jobs, clouds)
    # rows with exact ties (none on jittered synthetic clouds, most rows of a real scan): the reference's order, support stage by support stage
    trees = {}
    for j, (q_pts, s_pts, q_len, s_len, r, s_stage, t) in enumerate(searches):
        n_tie = int(words_host[n_jobs * n_clouds + j])
        if n_tie > 0:
            trees[s_stage] = _ops.radius_tie_order(jobs[j][1][0], q_pts, s_pts, q_len, s_len, r, t[0], n_tie, max(int(counts[j].max()), 1),
                                                   tree=trees.get(s_stage))
    if trees:
        _ops.tie_overflow_check()              # (one more small copy on the path that already copied the stage's points: a row left behind raises)
    num_pairs = counts.shape[1] // 2
    pair_counts = counts.view(counts.shape[0], num_pairs, 2).amax(2).tolist() if counts.shape[1] % 2 == 0 else None
    out = {'points': points_list, 'lengths': lengths_list, 'neighbors': [], 'subsampling': [], 'upsampling': []}
    stage_of = {'neighbors': lambda k: k, 'subsampling': lambda k: k + 1, 'upsampling': lambda k: k}
    for j, (kind, (full, _)) in enumerate(jobs):
        width = min(full.shape[1], int(counts[j].max()))
        if pair_counts is not None and 1 < num_pairs <= 64 and min(pair_counts[j]) < width and full.is_cuda:
            # several pairs stacked: a pair processed alone would have kept only min(limit, ITS max count) columns; columns
            # beyond that are marked -1 (ignored by every consumer, unlike the padding index Ns which selects the zero row): one launch
            q_lengths = lengths_list[stage_of[kind](len(out[kind]))]
            ends, row = [], 0
            for p in range(num_pairs):
                row += int(q_lengths[2 * p] + q_lengths[2 * p + 1])
                ends.append(row)
            table = _ops.neighbor_table_trim(full, width, ends, pair_counts[j])
        else:
            table = full if width == full.shape[1] else full[:, :width].contiguous()
            if pair_counts is not None and num_pairs > 1 and min(pair_counts[j]) < width:
                q_lengths = lengths_list[stage_of[kind](len(out[kind]))]
                row = 0
                for p in range(num_pairs):
                    rows_p = int(q_lengths[2 * p] + q_lengths[2 * p + 1])
                    if pair_counts[j][p] < width:
                        if table is full:
                            table = full.clone()
                        table[row:row + rows_p, pair_counts[j][p]:] = -1
                    row += rows_p
        out[kind].append(table)
    return out


def registration_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits,
                                       precompute_data=True, device='cuda'):
    """One-pair version of the reference collate: uploads the pair and builds the pyramid on the device."""
    if len(data_dicts) != 1:
        raise NotImplementedError('one registration pair per call (as the reference, batch_size = 1)')
    d = data_dicts[0]
    as_t = lambda a: torch.as_tensor(a)
    ref, src = as_t(d['ref_points']).float(), as_t(d['src_points']).float()
    out = {k: as_t(v).to(device) for k, v in d.items() if k not in ('ref_points', 'src_points', 'ref_feats', 'src_feats')}
    out['features'] = torch.cat((as_t(d['ref_feats']).float(), as_t(d['src_feats']).float()), 0).to(device)
    points = torch.cat((ref, src), 0).to(device)
    lengths = torch.tensor([ref.shape[0], src.shape[0]], dtype=torch.int64)
    if precompute_data:
        out.update(precompute_data_stack_mode(points, lengths, num_stages, voxel_size, search_radius, neighbor_limits))
    else:
        out['points'], out['lengths'] = points, lengths
    out['batch_size'] = 1
    return out
