"""Picture of a model and its synthetic logs: what the reference draws at the end of `save_results` (remo3d.py:993-1147) - the
formation as resistivity-coloured polygons (dipping layers, invasion zones, the borehole with its caliper) beside one or more
tracks of apparent-resistivity logs, saved as `Results_plot.png`.  Same keywords and defaults as the reference; unlike it this
module neither changes `Model.logs` (the reference overwrites them with the smoothed curves) nor the formation table (the reference
stretches its first and last row in place).  matplotlib is imported on first use, with a non-interactive backend if none is set."""
import os

import numpy as np


def _limits_of_logs(curves):
    """The reference's automatic resistivity range (remo3d.py:1015-1024): the extremes of everything in the log tables, rounded
    outwards to two significant digits of the maximum.  (Its scan runs over whole tables, depth column included; kept.)"""
    hi = max(float(np.nanmax(c)) for c in curves)
    lo = min(float(np.nanmin(c)) for c in curves)
    unit = 10.0 ** (np.floor(np.log10(hi)) - 1)
    return [float(np.floor(lo / unit) * unit), float(np.ceil(hi / unit) * unit)]


def model_polygons(formation, borehole, dip_deg, depth_lim, rad_lim):
    """(polygons, resistivities): one quadrilateral per layer spanning the picture, sheared by the dip; on top of it the invaded
    zone of a layer that has one (columns of the formation table: top, bottom, invasion radius, flushed-zone and virgin
    resistivity - remo3d.py:344-548); last the borehole between the mirrored caliper curves, coloured by its mean mud resistivity.
    The first and last layer are extended so that the sheared picture is filled (remo3d.py:1031-1032)."""
    f = np.array(formation, dtype=float, copy=True)
    slope = np.tan(np.deg2rad(dip_deg))
    f[0, 0] -= slope * rad_lim[1]
    f[-1, 1] += slope * rad_lim[1]
    polys, res = [], []

    def sheared(r0, r1, top, bottom):
        return np.array([[r0, top + slope * r0], [r0, bottom + slope * r0], [r1, bottom + slope * r1], [r1, top + slope * r1]])
    for top, bottom, r_inv, *rho in f:
        rho = [v for v in rho if not np.isnan(v)]
        polys.append(sheared(rad_lim[0], rad_lim[1], top, bottom))
        res.append(rho[-1])                                   # virgin zone (the table's last resistivity)
        if not np.isnan(r_inv):
            polys.append(sheared(-r_inv, r_inv, top, bottom))
            res.append(rho[0])                                # flushed zone
    if borehole is not None:
        b = np.asarray(borehole, dtype=float)
        polys.append(np.vstack([np.column_stack([-b[:, 1], b[:, 0]]), np.column_stack([b[:, 1], b[:, 0]])[::-1]]))
        res.append(float(np.mean(b[:, 2])))
    return polys, np.asarray(res)


def smoothed(log, factor):
    """Cubic resampling of one (depth, value) table to `factor` times its points (display only)."""
    if factor <= 1:
        return log
    from scipy.interpolate import interp1d
    z = np.linspace(log[:, 0].min(), log[:, 0].max(), int(log.shape[0] * factor))
    return np.column_stack([z, interp1d(log[:, 0], log[:, 1], kind="cubic")(z)])


def plot_results(model, path=None, plot_layout="auto", plot_depth_lim="auto", plot_aspect_ratio="auto", model_rad_lim="auto",
                 model_res_lim="auto", logs_res_lim="auto", logs_at_nan="break", logs_interpolation_factor=1, logs_colours="auto"):
    """Draws `model` (a remo3d_amd.Model with logs) and returns the matplotlib figure; `path`: also written there as PNG."""
    if logs_at_nan not in ("break", "continue"):
        raise ValueError('logs_at_nan paramater has to be set to "break" or "continue"')
    import matplotlib
    if not os.environ.get("MPLBACKEND") and not os.environ.get("DISPLAY"):
        matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    from matplotlib import ticker
    from matplotlib.collections import PatchCollection
    from matplotlib.patches import Polygon

    formation, borehole = model.formation_model, model.borehole_model
    logs = {k: smoothed(np.asarray(v, dtype=float), logs_interpolation_factor) for k, v in model.logs.items()}
    if plot_depth_lim == "auto":
        plot_depth_lim = [float(np.nanmin(formation[:, :2])), float(np.nanmax(formation[:, :2]))]
    if model_rad_lim == "auto":
        if np.all(np.isnan(formation[:, 2])):
            reach = 10.0 * float(np.nanmax(borehole[:, 1]))
        else:
            reach = 2.0 * float(np.nanmax(formation[:, 2]))
        model_rad_lim = [-reach, reach]
    if logs_res_lim == "auto":
        logs_res_lim = _limits_of_logs(list(logs.values()))
    if plot_aspect_ratio == "auto":
        plot_aspect_ratio = (plot_depth_lim[1] - plot_depth_lim[0]) / 25.0 * 1.25
    tracks = [list(logs.keys())] if plot_layout == "auto" else [list(t) for t in plot_layout]

    polys, res = model_polygons(formation, borehole, model.dip_deg, plot_depth_lim, model_rad_lim)
    width = 5 + 5 * len(tracks)
    style = {"font.size": 14, "axes.labelsize": 14, "axes.titlesize": 14, "xtick.labelsize": 14, "ytick.labelsize": 14, "axes.titlepad": 14,
             "xtick.major.size": 10, "xtick.minor.size": 5, "ytick.major.size": 10, "ytick.minor.size": 5}
    with plt.rc_context(style):
        fig, axes = plt.subplots(1, 1 + len(tracks), sharey=True, figsize=[width, width * plot_aspect_ratio], facecolor="white")
        picture = PatchCollection([Polygon(p, closed=True) for p in polys], cmap=matplotlib.colormaps["viridis"])
        picture.set_array(res)
        if model_res_lim != "auto":
            picture.set_clim(model_res_lim)
        left = axes[0]
        left.add_collection(picture)
        left.plot([0, 0], plot_depth_lim, color="black")
        left.margins(x=0, y=0)
        left.set_xlim(model_rad_lim)
        left.set_ylim(plot_depth_lim)
        left.invert_yaxis()
        left.minorticks_on()
        left.set_title("Formation model\ndip = %s\N{DEGREE SIGN}\n" % model.dip_deg)
        left.set_xlabel("Radial distance [m]", labelpad=10)
        left.set_ylabel("Depth [m]", labelpad=10)
        marks = left.get_xticks()
        left.xaxis.set_major_locator(ticker.FixedLocator(marks))
        left.set_xticklabels(["%.2f" % abs(t) for t in marks])
        left.xaxis.set_ticks_position("top")
        left.xaxis.set_label_position("top")
        cycle = plt.rcParams["axes.prop_cycle"].by_key()["color"]
        for t, names in enumerate(tracks):
            colours = cycle if logs_colours == "auto" else logs_colours[t]
            base = axes[1 + t]
            for i, name in enumerate(names):
                ax = base if i == 0 else base.twiny()
                curve = logs[name]
                if logs_at_nan == "continue":
                    curve = curve[~np.isnan(curve[:, 1])]
                colour = colours[i % len(colours)]
                ax.plot(curve[:, 1], curve[:, 0], color=colour)
                ax.set_xlabel(name + "\n[ohmm]", color=colour, labelpad=-8)
                ax.spines["top"].set_color(colour)
                ax.spines["top"].set_position(("outward", i * 55 + 10))
                ax.set_xticks(logs_res_lim)
                ax.tick_params(axis="x", color=colour)
                ax.set_xlim(logs_res_lim)
                ax.xaxis.set_label_position("top")
                ax.xaxis.set_ticks_position("top")
            base.grid(True)
            base.margins(x=0, y=0)
        bar = fig.colorbar(picture, ax=axes, location="bottom", orientation="horizontal", pad=0.05, label="Resistivity [ohmm]",
                           shrink=min(1.0, plot_aspect_ratio))
        bar.ax.minorticks_on()
        if path is not None:
            fig.savefig(path, bbox_inches="tight")
    return fig
