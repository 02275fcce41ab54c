"""The one live call of the reference's loop-closure stage.

slam/loop_closing.py is a stub: every stage of GlobalMapManager is `pass` except `_loop_candidate_detection`, which issues
`self.dbow.query(kf.des, [], 1, -1)` (:13-15) and drops the result; `update` would raise on the `None` it unpacks (:31-35) and the
class is never constructed (slam/main.py).  What can be matched is therefore that query.  A loop candidate must not come from
the keyframe's own neighbourhood, so the query here is restricted - through the fork's target_inds argument (slam/bow_db.py:37-39)
- to the database entries outside the keyframe's local window; with an empty exclusion it is exactly the reference's call."""
from . import bow


def loop_candidate_detection(dbow, kf, exclude_frame_ids=(), min_score=0.0):
    """dbow: bow.Dbow (mirror of slam/bow_db.py).  kf: object with .id and .des.  exclude_frame_ids: frame ids of the local window
    (CovisibilityGraph.get_local_map(kf), slam/covisibility_graph.py:51-78) plus kf itself.
    Returns (frame id, score) of the best remaining entry, or (None, None) - the (Id, Score) slam/bow_db.py:41 reads."""
    skip = {dbow.frame2entry[f] for f in exclude_frame_ids if f in dbow.frame2entry}
    if kf.id in dbow.frame2entry:
        skip.add(dbow.frame2entry[kf.id])
    subset = [e for e in sorted(dbow.entry2frame) if e not in skip] if skip else []
    if skip and not subset:
        return None, None
    fid, score = dbow.query(kf, subset_inds=subset)
    if fid is None or score < min_score:
        return None, None
    return fid, score
