"""Small matplotlib helper shared by the optional plots (role of the reference's utilities/plots.py)."""


def set_axis(ax, axis="both"):
    from matplotlib.ticker import AutoMinorLocator

    ax.tick_params(axis=axis, which="major", direction="in", length=7, width=1.5, labelsize=16,
                   top=True, right=True)
    ax.tick_params(axis=axis, which="minor", direction="in", length=4, width=1.2, top=True, right=True)
    if axis in ("x", "both"):
        ax.xaxis.set_minor_locator(AutoMinorLocator(2))
    if axis in ("y", "both"):
        ax.yaxis.set_minor_locator(AutoMinorLocator(2))
    for side in ("top", "bottom", "left", "right"):
        ax.spines[side].set_linewidth(1.5)
